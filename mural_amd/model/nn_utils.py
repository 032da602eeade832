"""Model factory / initialiser / batched inference loop with the reference's contracts.

  * ``model_choice``    -- MuRaL/model/nn_utils.py:186-231: same 4 positional arguments, same config keys, same
                           registry ({'snv': {0,1,2}, 'indel': {0}}) and the same ValueError texts.
  * ``weights_init``    -- nn_utils.py:14-35 (xavier-uniform convs, kaiming-normal linears, zero biases), applied with
                           ``model.apply`` like training.py:324.
  * ``model_predict_m`` -- nn_utils.py:37-76: returns ``(pred_y (N, n_class) on device, total_loss float)``; the loss
                           is accumulated on the device and read back ONCE (the reference syncs per batch, :65), the
                           outputs are concatenated once (the reference re-copies the growing tensor per batch, :62) and
                           small loader batches are fused into one launch.
"""
import inspect
from operator import attrgetter

import numpy as np
import torch
import torch.nn as nn

from .model_snv import Network0, Network1, Network2

MODEL_REGISTRY = {"snv": {0: Network0, 1: Network1, 2: Network2}, "indel": {}}

try:  # the INDEL model lands in its own module
    from .model_indel import UNet_Small
    MODEL_REGISTRY["indel"][0] = UNet_Small
except ImportError:  # pragma: no cover
    pass


def weights_init(m):
    """Initialise network layers by class name, as the reference does."""
    classname = m.__class__.__name__
    if classname.find("Conv1d") != -1 or classname.find("Conv2d") != -1:
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif classname.find("Linear") != -1:
        nn.init.kaiming_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


def model_choice(model_no, config, common_model_config, model_type):
    """Build the model the config describes; kwargs are matched against the constructor signature."""
    model_config = {**config, **common_model_config}

    def adapt(para):
        if model_type == "snv":
            rules = {
                "lin_layer_sizes": lambda: [config["local_hidden1_size"], config["local_hidden2_size"]],
                "lin_layer_dropouts": lambda: [config["local_dropout"], config["local_dropout"]],
                "emb_padding_idx": lambda: 4 ** config["local_order"],
                "out_channels": lambda: config["CNN_out_channels"],
                "kernel_size": lambda: config["CNN_kernel_size"],
                "no_of_cont": lambda: common_model_config["n_cont"],
            }
        else:
            rules = {
                "out_channels": lambda: config["CNN_out_channels"],
                "kernel_size": lambda: config["CNN_kernel_size"],
                "downsize": lambda: config["down_list"],
                "use_reverse": lambda: config.get("use_reverse", False),
            }
        return rules[para]() if para in rules else model_config[para]

    if model_type not in MODEL_REGISTRY:
        raise ValueError(f"model_type must be one of {list(MODEL_REGISTRY.keys())}, got {model_type}")
    model_map = MODEL_REGISTRY[model_type]
    model = model_map.get(model_no)
    if model is None:
        raise ValueError(f"model_no for {model_type} must be one of {list(model_map.keys())}, got {model_no}")
    names = [p for p in inspect.signature(model.__init__).parameters.keys() if p != "self"]
    return model(**{p: adapt(p) for p in names})


def _gather_to_device(pending, k, device):
    """Field k of the waiting loader batches as ONE device tensor: every batch is copied to the device as it is (a 16-row batch is
    128 KB: the 512 copies of a flush move 262 MB in 17 ms) and the pieces are concatenated there.  Packing on the host first was
    10 - 40x slower on this pool's 256-CPU hosts: every torch CPU copy / cat enters the intra-op thread pool (154 ms for the same
    262 MB with the default thread count, 6 ms with 8 threads -- not something a library call should depend on)."""
    parts = [b[k] for b in pending]
    if len(parts) == 1:
        return parts[0].to(device, non_blocking=True)
    if not parts[0].is_cuda:
        parts = [p.to(device, non_blocking=True) for p in parts]
    return torch.cat(parts, dim=0).to(device)


_DTYPE, _IS_CUDA = attrgetter("dtype"), attrgetter("is_cuda")


class _HostSymbolRoute:
    """model_predict_m for a loader that yields HOST tensors (the reference's: nn_utils.py:52-56 copies y, cont_x, cat_x, distal_x to
    the device batch by batch -- 32 KB of fp32 one-hot per site over PCIe).  The waiting batches' windows are classified into one
    symbol byte per column by host threads (``mural_host_dense_to_symbols``, the host twin of the dense entry's first pass) straight
    into one of two pinned staging buffers, 1 / 16 of the bytes cross PCIe, and the launch takes the symbols.  Columns that are no
    MuRaL encoding raise ValueError at once (the dense entry reports them one call late)."""

    _staging = {}          # (rows, L) -> pinned buffers: pinning 16 MB costs milliseconds, a predict loop is called per file

    def __init__(self, model, device, fuse_rows):
        self.model, self.device, self.L = model, device, model.seq_len
        self.cap_rows = fuse_rows + 1024                  # a flush holds fuse_rows rows + what its last batch brought: pin once
        key = (fuse_rows, self.L)
        self.bufs = self._staging.setdefault(key, [None, None])
        self.small = self._staging.setdefault(("small",) + key, [None, None])
        self.events, self.turn = [None, None], 0

    @staticmethod
    def model_ok(model, model_type, distal):
        return model_type == "snv" and distal and getattr(model, "symbols_entry_ok", lambda: False)()

    _pool = None           # one helper thread: the host classification of flush k runs beside the collection of flush k + 1

    def begin(self, pending, rows):
        """Start the host side of a flush: checks and addresses here, classification + packing of the small fields on the helper thread.
        Returns a job for ``finish`` -- or None if a batch is not what the reference's loader yields (host fp32 (b, 4, L) windows, host
        int64 cat_x, host fp32 (b, 1) labels): the caller then takes the dense entry."""
        import ctypes as C
        from .. import _lib
        L, f32, i64, T = self.L, torch.float32, torch.int64, torch.Tensor
        # checks and addresses of ~512 batches per flush: this is the route's host cost, so every field is read by ONE C-level map over the
        # batches (a Python loop with a dozen attribute reads per batch took 2.6 ms per flush, as long as the classification beside it)
        ys, cs, xs = [b[0] for b in pending], [b[2] for b in pending], [b[3] for b in pending]
        if not all(map(torch.is_tensor, ys)) or not all(map(torch.is_tensor, cs)) or not all(map(torch.is_tensor, xs)):
            return None
        cols = cs[0].shape[1] if cs[0].dim() == 2 else -1
        sy = list(map(T.size, ys))
        ns = [s[0] if len(s) else -1 for s in sy]
        if (set(map(_DTYPE, xs)) != {f32} or set(map(_DTYPE, cs)) != {i64} or set(map(_DTYPE, ys)) != {f32}
                or any(map(_IS_CUDA, xs)) or any(map(_IS_CUDA, cs)) or any(map(_IS_CUDA, ys))
                or not (all(map(T.is_contiguous, xs)) and all(map(T.is_contiguous, cs)) and all(map(T.is_contiguous, ys)))
                or list(map(T.size, xs)) != [(n, 4, L) for n in ns] or list(map(T.size, cs)) != [(n, cols) for n in ns]
                or sy != [(n, 1) for n in ns]):
            return None
        i = self.turn
        self.turn ^= 1
        if self.bufs[i] is None or self.bufs[i].numel() < rows * L:
            self.bufs[i] = torch.empty(max(rows, self.cap_rows) * L, dtype=torch.uint8).pin_memory()
        need = rows * (8 * cols + 4)
        if self.small[i] is None or self.small[i].numel() < need:
            self.small[i] = torch.empty(max(rows, self.cap_rows) * (8 * cols + 4), dtype=torch.uint8).pin_memory()
        if self.events[i] is not None:
            self.events[i].synchronize()               # the copies out of these buffers two flushes ago
        nb = len(pending)
        lib = _lib.lib()
        sym_h, small = self.bufs[i], self.small[i]
        n64 = np.array(ns, dtype=np.int64)
        args = (np.fromiter(map(T.data_ptr, xs), dtype=np.int64, count=nb), n64,
                np.fromiter(map(T.data_ptr, cs), dtype=np.int64, count=nb), n64 * (8 * cols),
                np.fromiter(map(T.data_ptr, ys), dtype=np.int64, count=nb), n64 * 4)
        args = tuple(a.ctypes.data_as(C.c_void_p) for a in args) + (args,)      # (addresses of the arrays; the arrays themselves stay alive)

        def host_work():
            bad = C.c_int64(0)
            _lib.check(lib.mural_host_dense_to_symbols(args[0], args[1], nb, L, sym_h.data_ptr(), C.byref(bad)))
            _lib.check(lib.mural_host_concat(args[2], args[3], nb, small.data_ptr()))
            _lib.check(lib.mural_host_concat(args[4], args[5], nb, small.data_ptr() + rows * 8 * cols))
            return bad.value

        if _HostSymbolRoute._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _HostSymbolRoute._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mural-host-classify")
        return (_HostSymbolRoute._pool.submit(host_work), i, rows, cols, pending)      # (pending: the batches stay alive until read)

    def finish(self, job):
        """(symbols, cat_x, y) on the device for a job of ``begin``."""
        fut, i, rows, cols, _ = job
        if fut.result():
            raise ValueError(type(self.model)._ENC_MSG)
        L, need = self.L, rows * (8 * cols + 4)
        sym_h, small = self.bufs[i], self.small[i]
        sym = sym_h[:rows * L].view(rows, L).to(self.device, non_blocking=True)
        cat_x = small[:rows * 8 * cols].view(torch.int64).view(rows, cols).to(self.device, non_blocking=True)
        y = small[rows * 8 * cols:need].view(torch.float32).view(rows, 1).to(self.device, non_blocking=True)
        self.events[i] = torch.cuda.Event()
        self.events[i].record()
        return sym, cat_x, y

    def gather(self, pending, rows):
        job = self.begin(pending, rows)
        return None if job is None else self.finish(job)


def model_predict_m(model, dataloader, criterion, device, n_class, distal=True, model_type="snv", fuse_rows=8192):
    """Run the model over an iterable of (y, cont_x, cat_x, distal_x) batches.

    The reference launches one forward per loader batch (default 16 rows, commands/predict.py:90); a 16-row launch leaves
    the GPU idle, so consecutive loader batches are concatenated until `fuse_rows` rows are waiting and evaluated by ONE
    forward.  Results are unchanged: rows keep their order, eval-mode outputs do not depend on the batch they are computed in,
    and the loss is the sum of the per-batch criterion values (taken on the slices when the criterion does not sum)."""
    from .._host import freeze_host_heap
    freeze_host_heap()
    device = torch.device(device)
    model.to(device)
    model.eval()
    outs = []
    loss_acc = torch.zeros((), dtype=torch.float64, device=device)
    additive = getattr(criterion, "reduction", None) == "sum"
    pending, rows = [], 0

    host_route, host_ok = None, _HostSymbolRoute.model_ok(model, model_type, distal)
    in_flight = None       # a host-route flush whose classification runs on the helper thread: (job, batches)

    def account(preds, y, batches):
        outs.append(preds)
        target = y.long().squeeze(1)
        if additive or len(batches) == 1:
            loss_acc.add_(criterion(preds, target).double())
        else:
            o = 0
            for b in batches:
                n = b[0].shape[0]
                loss_acc.add_(criterion(preds[o:o + n], target[o:o + n]).double())
                o += n

    def complete():
        nonlocal in_flight
        if in_flight is None:
            return
        job, batches = in_flight
        in_flight = None
        with torch.cuda.device(device):
            sym, cat_x, y = host_route.finish(job)
            preds = model.forward_symbols(cat_x, sym)
        account(preds, y, batches)

    def flush():
        nonlocal pending, rows, host_route, in_flight
        if not pending:
            return
        job = None
        if host_ok and not pending[0][3].is_cuda:
            if host_route is None:
                host_route = _HostSymbolRoute(model, device, fuse_rows)
            with torch.cuda.device(device):
                job = host_route.begin(pending, rows)      # the windows are classified on a helper thread while ...
        complete()                                         # ... the previous flush is uploaded and launched, and the next one collected
        if job is not None:
            in_flight = (job, pending)
        else:
            y, cont_x, cat_x, distal_x = (_gather_to_device(pending, k, device) for k in range(4))
            if model_type == "snv":
                preds = model.forward((cont_x, cat_x), distal_x) if distal else model.forward(cont_x, cat_x)
            else:
                preds = model.forward(distal_x)
            account(preds, y, pending)
        pending, rows = [], 0

    # the first flushes are short (fuse_rows / 4, / 2, then fuse_rows): the helper thread, the copy engine and the GPU start working after
    # a quarter of a flush's collection time instead of a whole one (a loader of two flushes' worth of rows otherwise spends half its time
    # before anything overlaps)
    flush_at = max(fuse_rows // 4, 1)
    with torch.no_grad():
        try:
            for batch in dataloader:
                pending.append(batch)
                rows += batch[0].shape[0]
                if rows >= flush_at:
                    flush()
                    flush_at = min(2 * flush_at, fuse_rows)
            flush()
            complete()
        finally:
            if in_flight is not None:      # an error on the way: do not leave the helper thread writing into staging that will be reused
                try:
                    in_flight[0][0].result()
                except Exception:      # noqa: BLE001
                    pass
    check = getattr(getattr(model, "model", model), "check_encoding", None)     # Network0 wraps its body in .model
    if check is not None:
        check(wait=True)
    pred_y = torch.cat(outs, dim=0) if outs else torch.empty(0, n_class, device=device)
    return pred_y, float(loss_acc.item())


# ------------------------------------------------------------------------------------------------------------------
# model directories written by the reference's training run (`model`, `model.config.pkl`, `model.fdiri_cal.pkl`)
# ------------------------------------------------------------------------------------------------------------------
class _ConfigUnpickler(__import__("pickle").Unpickler):
    """``model.config.pkl`` is a plain dict of Python / numpy scalars, lists and tuples: anything else is refused."""

    _OK = {("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
           ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
           ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        import pickle
        raise pickle.UnpicklingError(f"model config references {module}.{name}: refused")


def load_model_config(config_path):
    """The hyper-parameter dict the reference pickles next to every checkpoint (scripts/run_predict.py:58-64)."""
    with open(config_path, "rb") as fh:
        config = _ConfigUnpickler(fh).load()
    if not isinstance(config, dict):
        raise ValueError(f"{config_path}: not a model config (expected a dict, got {type(config).__name__})")
    return config


def load_model(model_path, config_path=None, model_type="snv", device="cuda"):
    """Build the network a checkpoint was trained with and load its weights, as scripts/run_predict.py:58-91, :163-189 does:
    ``model_choice(config['model_no'], config, {'emb_dims', 'n_cont': 0, 'n_class', 'distal_order': 1, 'in_channels': 4})``
    then ``load_state_dict``.  ``config_path`` defaults to ``<model_path>.config.pkl``.  Sequence-only models (every shipped
    one: ``seq_only``) are supported; bigWig covariates are out of scope.  Returns (model in eval mode on `device`, config)."""
    import torch
    config = load_model_config(config_path or model_path + ".config.pkl")
    if not config.get("seq_only", True):
        raise ValueError("this checkpoint uses bigWig covariates (seq_only=False): not supported by the HIP path")
    common = {"emb_dims": config["emb_dims"], "n_cont": 0, "n_class": config["n_class"], "distal_order": 1, "in_channels": 4}
    model = model_choice(config["model_no"], config, common, model_type)
    state = torch.load(model_path, map_location="cpu", weights_only=True)
    model.load_state_dict(state)
    return model.to(device).eval(), config


def save_model(model, dirichlet_weights, config, save_path):
    """The three files of MuRaL/training.py:570-578: ``save_path`` (state dict), ``save_path + '.fdiri_cal.pkl'`` (skipped when
    `dirichlet_weights` is None) and ``save_path + '.config.pkl'`` -- loadable by the reference and by ``load_model`` /
    ``calibration.load_dirichlet_weights``."""
    import pickle

    import torch
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, save_path)
    if dirichlet_weights is not None:
        from ..calibration import save_dirichlet_calibrator
        save_dirichlet_calibrator(dirichlet_weights, save_path + ".fdiri_cal.pkl")
    with open(save_path + ".config.pkl", "wb") as fp:
        pickle.dump(dict(config), fp)
