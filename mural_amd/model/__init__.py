from .nn_utils import model_choice, model_predict_m, weights_init  # noqa: F401
