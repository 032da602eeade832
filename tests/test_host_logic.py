"""CPU tests of the host-side logic: model factory / state-dict layout, C-ABI symbols, row ordering, batching,
genome packing, shard partition.  No compute on a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests import _util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header_name):
    header = open(os.path.join(ROOT, "include", header_name)).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)        # symbols named in comments are not declarations
    return set(re.findall(r"\b(mural_[a-z0-9_]+)\s*\(", header))


def test_abi_library_exports_every_declared_symbol():
    """Both flavours export every symbol of include/mural_hip.h; the product library exports NO validation hook, the debug flavour all of
    include/mural_hip_debug.h; the ctypes prototypes cover exactly the declarations."""
    import subprocess
    from mural_amd import _lib
    declared, hooks = _declared("mural_hip.h"), _declared("mural_hip_debug.h")
    assert declared and hooks and not any(n.startswith("mural_debug_") for n in declared)
    assert all(n.startswith("mural_debug_") for n in hooks)
    product, debug = ctypes.CDLL(_lib.LIB_PATH), ctypes.CDLL(_lib.DEBUG_LIB_PATH)
    for name in sorted(declared):
        assert hasattr(product, name), f"{name} is declared in include/mural_hip.h but not exported"
        assert hasattr(debug, name)
    for name in sorted(hooks):
        assert hasattr(debug, name), f"{name} is declared in include/mural_hip_debug.h but not exported by the debug flavour"
        assert not hasattr(product, name), f"the product library exports the validation hook {name}"
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "mural_debug" not in exported
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert hooks == set(_lib.DEBUG_PROTOTYPES), hooks ^ set(_lib.DEBUG_PROTOTYPES)
    assert _lib.lib().mural_abi_version() >= 1


def test_every_development_switch_is_in_the_one_table():
    """The product library reads MURAL_HOST_THREADS and TMPDIR; every other environment switch of csrc/ goes through dev_env() and is
    listed in dev_switch_table (one line of description each), which only the debug flavour honours."""
    import glob
    from mural_amd import _lib
    assert _lib.flavor() == "debug"
    buf = ctypes.create_string_buffer(1 << 16)
    n = _lib.lib().mural_debug_list_switches(buf, len(buf))
    table = dict(ln.split("\t", 1) for ln in buf.value.decode().splitlines())
    assert len(table) == n and all(len(v) > 10 for v in table.values())
    used, raw = set(), set()
    for path in glob.glob(os.path.join(ROOT, "mural_amd", "csrc", "*.h*")):
        src = open(path).read()
        used |= set(re.findall(r'dev_env\("([A-Z_0-9]+)"\)', src))
        raw |= set(re.findall(r'(?<![a-z_])getenv\("([A-Z_0-9]+)"\)', src))
    assert used == set(table), used ^ set(table)
    assert raw == {"MURAL_HOST_THREADS", "TMPDIR"}, raw


def _cfg(r=7, R=1000, n_class=4):
    ncol = 2 * r + 1 - 2
    cfg = dict(local_radius=r, local_order=3, local_hidden1_size=150, local_hidden2_size=75, distal_radius=R,
               emb_dropout=0.1, local_dropout=0.1, CNN_kernel_size=3, CNN_out_channels=32, distal_fc_dropout=0.25)
    common = dict(emb_dims=[(65, 2)] * ncol, n_cont=0, n_class=n_class, distal_order=1, in_channels=4)
    return cfg, common


def test_model_choice_matches_reference_state_dict_layout():
    from mural_amd.model import model_choice
    cfg, common = _cfg()
    model = model_choice(2, cfg, common, "snv")
    fx = U.load("snv_pretrained_human_AT.npz")
    oracle = U.snv_oracle_from_hp(fx["hp"])
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    assert len(model.state_dict()) == 302 and sum(p.numel() for p in model.parameters()) == 86904
    # the shipped checkpoint loads strictly (run_predict.py:188)
    model.load_state_dict(U.expand_state(model.state_dict().keys(), U.fixture_weights(fx)), strict=True)
    for no in (0, 1):
        m = model_choice(no, cfg, common, "snv")
        o = U.snv_oracle_from_hp(np.array([7, 3, 1000, 150, 75, 32, 3, 4, no]))
        assert list(m.state_dict().keys()) == list(o.state_dict().keys())


def test_model_choice_errors_like_the_reference():
    from mural_amd.model import model_choice
    cfg, common = _cfg()
    with pytest.raises(ValueError, match="model_type must be one of"):
        model_choice(2, cfg, common, "sv")
    with pytest.raises(ValueError, match="model_no for snv must be one of"):
        model_choice(3, cfg, common, "snv")


def test_weights_init_by_class_name():
    from mural_amd.model import model_choice, weights_init
    cfg, common = _cfg(5, 100)
    torch.manual_seed(0)
    model = model_choice(2, cfg, common, "snv")
    emb_before = model.emb_layer.weight.clone()
    model.apply(weights_init)
    assert torch.equal(model.emb_layer.weight, emb_before)            # embeddings keep torch's default init
    assert float(model.conv1[1].bias.abs().max()) == 0.0 and float(model.lin_layers[0].bias.abs().max()) == 0.0
    w = model.RBs1[0].conv1.weight
    bound = (6.0 / (32 * 3 + 32 * 3)) ** 0.5                             # xavier_uniform
    assert float(w.abs().max()) <= bound + 1e-6


def test_no_cpu_fallback():
    from mural_amd.model import model_choice
    cfg, common = _cfg(5, 100)
    model = model_choice(2, cfg, common, "snv").eval()
    with pytest.raises(RuntimeError, match="HIP device"):
        model((torch.zeros(2, 1), torch.zeros(2, 9, dtype=torch.long)), torch.zeros(2, 4, 201))
    model.train()
    with pytest.raises(RuntimeError, match="HIP device"):
        model((torch.zeros(2, 1), torch.zeros(2, 9, dtype=torch.long)), torch.zeros(2, 4, 201))


def test_segment_order_matches_bed_reader():
    from mural_amd.data.batching import segment_order
    fx = U.load("windowing.npz")
    order, group = segment_order(fx["in_chrom"], fx["in_start"], fx["in_strand"], int(fx["central"]))
    assert np.array_equal(fx["in_chrom"][order], fx["out_chrom"])
    assert np.array_equal(fx["in_start"][order], fx["out_start"])
    assert np.array_equal(fx["in_strand"][order], fx["out_strand"])
    assert np.array_equal(group, fx["out_group"])


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_generate_data_batches_row_order(case):
    from mural_amd.data.batching import generate_data_batches
    fx = U.load("batching.npz")
    bs, nseg = [int(v) for v in fx[f"{case}_bs_nseg"]]
    segs, o = [], 0
    for n in fx[f"{case}_sizes"].tolist():
        ids = torch.arange(o, o + n, dtype=torch.float32).reshape(1, n, 1)
        segs.append((ids, torch.zeros(1, n, 1), ids.long().repeat(1, 1, 3), ids.reshape(1, n, 1, 1).repeat(1, 1, 4, 6)))
        o += n
    rows, cuts = [], []
    for y, cont, cat, dist in generate_data_batches(segs, nseg, bs, shuffle=False):
        rows.extend(y[:, 0].long().tolist())
        cuts.append(y.shape[0])
        assert cont.dtype == torch.float64 and cont.shape == (y.shape[0], 1)
        assert torch.equal(cat[:, 0], y[:, 0].long()) and torch.equal(dist[:, 0, 0], y[:, 0])
    assert rows == fx[f"{case}_rows"].tolist()
    assert cuts == fx[f"{case}_cuts"].tolist()


def test_pack_sequence_matches_oracle_format():
    from mural_amd.data.genome import pack_sequence
    from oracle import encode_ref
    fx = U.load("encode.npz")
    seq = fx["seq"].tobytes().decode()
    packed, mask, n, amb = pack_sequence(seq)
    codes = encode_ref.seq_to_codes(seq)
    p2, m2 = encode_ref.pack_codes(codes)
    assert n == len(seq) and np.array_equal(packed, p2) and np.array_equal(mask, m2)
    assert np.array_equal(amb[0], np.nonzero(codes > 4)[0]) and np.array_equal(amb[1], codes[codes > 4])
    with pytest.raises(KeyError):
        pack_sequence("ACGTX")


def test_shard_bounds_partition():
    from mural_amd.predict import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- model directories (model, model.config.pkl) -------------------------------------------------------------------------
def test_load_model_from_reference_style_directory(tmp_path):
    import pickle
    from mural_amd.model import model_choice, nn_utils
    r, order = 4, 3
    ncol = 2 * r + 1 - (order - 1)
    config = dict(local_radius=r, local_order=order, local_hidden1_size=150, local_hidden2_size=75, distal_radius=300, emb_dropout=0.1,
                  local_dropout=0.1, CNN_kernel_size=3, CNN_out_channels=32, distal_fc_dropout=0.25, n_class=4, model_no=2,
                  seq_only=True, emb_dims=[(np.int64(4 ** order + 1), 2)] * ncol, segment_center=300000)
    common = dict(emb_dims=config["emb_dims"], n_cont=0, n_class=4, distal_order=1, in_channels=4)
    torch.manual_seed(3)
    src = model_choice(2, config, common, "snv")
    torch.save(src.state_dict(), tmp_path / "model")
    with open(tmp_path / "model.config.pkl", "wb") as fh:
        pickle.dump(config, fh)
    model, cfg = nn_utils.load_model(str(tmp_path / "model"), device="cpu")
    assert cfg["local_radius"] == r and not model.training and type(model).__name__ == "Network2"
    # save_model writes the reference's three files back (training.py:570-578)
    from mural_amd.calibration import load_dirichlet_weights
    w = np.hstack([np.eye(4), np.zeros((4, 1))])
    nn_utils.save_model(model, w, cfg, str(tmp_path / "ckpt"))
    again, cfg2 = nn_utils.load_model(str(tmp_path / "ckpt"), device="cpu")
    assert cfg2 == cfg and np.array_equal(load_dirichlet_weights(str(tmp_path / "ckpt.fdiri_cal.pkl")), w)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), again.state_dict().values()))
    for (k, a), (k2, b) in zip(src.state_dict().items(), model.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    # anything but plain data in the config pickle is refused
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("true",))
    with open(tmp_path / "bad.pkl", "wb") as fh:
        pickle.dump({"x": Evil()}, fh)
    with pytest.raises(pickle.UnpicklingError):
        nn_utils.load_model_config(str(tmp_path / "bad.pkl"))
    config["seq_only"] = False
    with open(tmp_path / "model.config.pkl", "wb") as fh:
        pickle.dump(config, fh)
    with pytest.raises(ValueError, match="bigWig"):
        nn_utils.load_model(str(tmp_path / "model"), device="cpu")


def test_load_shipped_checkpoints_if_present():
    from oracle import ref_import
    from mural_amd.model import nn_utils
    root = os.path.join(ref_import.REFERENCE_ROOT, "models", "Homo_sapiens")
    if not os.path.isdir(root):
        pytest.skip("reference tree not mounted")
    m, cfg = nn_utils.load_model(os.path.join(root, "SNV", "AT", "model"), device="cpu")
    assert cfg["model_no"] == 2 and len(m.state_dict()) == 302
    m, cfg = nn_utils.load_model(os.path.join(root, "INDEL", "insertion", "model"), model_type="indel", device="cpu")
    assert cfg["use_reverse"] and len(m.state_dict()) == 232


def test_train_epoch_policy_matches_reference_loop():
    """Host policy of MuRaL/training.py:392-450 on a stand-in model: single-row batches are skipped, the scheduler steps every
    batch, a learning rate below min_lr restarts at restart_lr, the loss is summed."""
    from mural_amd import train as TR

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 4)

        def forward(self, local, distal):
            return self.lin(distal.mean(dim=2)[:, :3])

    torch.manual_seed(0)
    model = Toy()
    config = dict(optim="Adam", learning_rate=1e-3, weight_decay=0.0, lr_scheduler="StepLR", batch_size=320000, LR_gamma=0.5,
                  min_lr=3e-4, restart_lr=8e-4)
    opt = TR.make_optimizer(config, model.parameters())
    sch = TR.make_scheduler(config, opt)
    assert sch.step_size == 2
    mk = lambda b: (torch.zeros(b, 1), torch.zeros(b, 1, dtype=torch.float64), torch.zeros(b, 9, dtype=torch.int64), torch.rand(b, 4, 21))  # noqa: E731
    batches = [mk(4), mk(1), mk(5), mk(4), mk(4), mk(4)]
    before = model.lin.weight.detach().clone()
    total = TR.train_epoch(model, batches, torch.nn.CrossEntropyLoss(reduction="sum"), opt, sch, config, "cpu")
    assert total > 0 and not torch.equal(before, model.lin.weight)
    # 5 batches trained (the single-row one skipped): lr 1e-3 -> 5e-4 after 2 steps -> 2.5e-4 < min_lr after 4 -> restart 8e-4
    assert opt.param_groups[0]["lr"] == pytest.approx(8e-4)
    with pytest.raises(ValueError):
        TR.make_optimizer(dict(config, optim="LBFGS"), model.parameters())
    assert isinstance(TR.make_optimizer(dict(config, optim="AdamW2"), model.parameters()), torch.optim.AdamW)


def test_cross_entropy_sum_takes_torch_path_off_device_and_symbol_windows_are_typed():
    """mural_amd.train.CrossEntropySum is torch's CrossEntropyLoss(reduction='sum') for anything but a float32 device tensor (the fused
    launch is the GPU tests' matter); SymbolWindows is what encode_symbols hands to the training forward, and a dense model call with it in
    eval mode, or with a malformed tensor inside, is refused before anything reaches the device."""
    import pytest
    import torch
    from mural_amd.train import CrossEntropySum
    from mural_amd.data import SymbolWindows
    from mural_amd.model import model_snv
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 4, generator=g, requires_grad=True)
    y = torch.randint(0, 4, (37,), generator=g)
    a = CrossEntropySum()(x, y)
    b = torch.nn.CrossEntropyLoss(reduction="sum")(x, y)
    assert torch.equal(a, b)
    a.backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()

    class Stub:                                   # what _symbol_windows looks at
        seq_len, training, in_channels = 201, False, 4
    ok = SymbolWindows(torch.zeros((3, 201), dtype=torch.uint8))
    assert tuple(ok.shape) == (3, 201)
    assert model_snv._symbol_windows(Stub(), torch.zeros(3, 4, 201)) is None          # a dense tensor takes the dense route
    with pytest.raises(TypeError):
        model_snv._symbol_windows(Stub(), ok)                                          # eval mode: forward_packed is the packed entry
    with pytest.raises(TypeError):
        model_snv._symbol_windows(Stub(), SymbolWindows(torch.zeros((3, 201), dtype=torch.float32)))
    with pytest.raises(ValueError):
        model_snv._symbol_windows(Stub(), SymbolWindows(torch.zeros((3, 301), dtype=torch.uint8)))


def test_host_dense_to_symbols_matches_the_encoding_table():
    """mural_host_dense_to_symbols (the host twin of the dense entry's first pass, used by model_predict_m for host loaders): every
    MuRaL column pattern -> its symbol, anything else -> 255 and counted; batches of unequal size, windows across thread borders."""
    import ctypes as C
    import numpy as np
    import torch
    from mural_amd import _lib
    from tests import _util as U
    rng = np.random.default_rng(0)
    L = 333
    codes = rng.integers(0, 15, size=(700, L)).astype(np.uint8)
    x = U.onehot(codes)
    sizes = [1, 16, 300, 7, 376]
    batches, o = [], 0
    for n in sizes:
        batches.append(x[o:o + n].contiguous())
        o += n
    ptrs = (C.c_void_p * len(batches))(*[b.data_ptr() for b in batches])
    counts = (C.c_int64 * len(batches))(*sizes)
    out = np.full((700, L), 77, np.uint8)
    bad = C.c_int64(-1)
    _lib.check(_lib.lib().mural_host_dense_to_symbols(ptrs, counts, len(batches), L, out.ctypes.data, C.byref(bad)))
    assert bad.value == 0 and np.array_equal(out, codes)
    batches[2][5, 1, 17] = 0.3                       # not a fraction of the encoding
    batches[4][0, :, 0] = torch.tensor([1.0, 1.0, 0.0, 0.0])      # two ones: no symbol has that column
    _lib.check(_lib.lib().mural_host_dense_to_symbols(ptrs, counts, len(batches), L, out.ctypes.data, C.byref(bad)))
    assert bad.value == 2 and out[17 + 5, 17] == 255 and out[324, 0] == 255
    assert (out != codes).sum() == 2
    # blocks of 256 columns that hold nothing but 0.0 and 1.0 take the bit-compare path of csrc/host_classify.cpp, any other block the
    # digit rule: A C G T windows with one run of N, a -0.0 (a zero of the rule, not of the bit compare), a column of two ones, a column of zeros
    L2 = 777
    codes2 = rng.integers(0, 4, size=(40, L2)).astype(np.uint8)
    codes2[3, 300:340] = 4                                   # N: four 0.25
    x2 = U.onehot(codes2).contiguous()
    x2[5, 2, 600] = -0.0 if codes2[5, 600] != 2 else x2[5, 2, 600]
    x2[7, :, 10] = torch.tensor([0.0, 1.0, 0.0, 1.0])
    x2[8, :, 770] = 0.0
    ptr2 = (C.c_void_p * 1)(x2.data_ptr())
    cnt2 = (C.c_int64 * 1)(40)
    out2 = np.full((40, L2), 77, np.uint8)
    _lib.check(_lib.lib().mural_host_dense_to_symbols(ptr2, cnt2, 1, L2, out2.ctypes.data, C.byref(bad)))
    want2 = codes2.copy()
    want2[7, 10] = want2[8, 770] = 255
    assert bad.value == 2 and np.array_equal(out2, want2)


def test_train_adam_is_torch_adam_off_the_hip_path():
    """mural_amd.train.Adam on parameters that are no HIP model's (here: CPU tensors) is torch.optim.Adam itself -- same constructor,
    bitwise the same steps, the same state_dict -- and make_optimizer hands it out for config['optim'] == 'Adam' (training.py:346-350)."""
    import torch
    from mural_amd.train import Adam, make_optimizer
    torch.manual_seed(0)
    a = torch.nn.Linear(5, 3)
    b = torch.nn.Linear(5, 3)
    b.load_state_dict(a.state_dict())
    oa, ob = Adam(a.parameters(), lr=1e-2, weight_decay=1e-3), torch.optim.Adam(b.parameters(), lr=1e-2, weight_decay=1e-3)
    for _ in range(3):
        x = torch.randn(7, 5)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).square().sum().backward()
            o.step()
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"] and sa["state"].keys() == sb["state"].keys()
    assert all(torch.equal(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"]) and float(sa["state"][k]["step"]) == 3.0 for k in sa["state"])
    oa.load_state_dict(sb)
    assert oa.step(lambda: torch.tensor(1.0)) == torch.tensor(1.0)      # a closure goes to torch's step as well
    seen = []
    h = oa.register_step_post_hook(lambda o, args, kwargs: seen.append(1))      # step hooks keep firing (they hang off torch's step)
    oa.zero_grad()
    a(torch.randn(2, 5)).sum().backward()
    oa.step()
    h.remove()
    assert seen == [1]
    opt = make_optimizer({"optim": "Adam", "learning_rate": 1e-3, "weight_decay": 1e-5}, a.parameters())
    assert isinstance(opt, Adam) and isinstance(opt, torch.optim.Adam)
