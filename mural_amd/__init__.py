"""mural_amd -- MI355X (gfx950) native hot path for MuRaL mutation-rate models.

Public surface mirrors the reference's seam (MuRaL/model/nn_utils.py): ``model_choice``, ``weights_init``,
``model_predict_m``; plus the packed-genome encoders and the sharded predictor this build adds.
"""
__version__ = "0.1.0"
