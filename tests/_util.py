"""Shared helpers for the parity tests (fixture loading, oracle construction)."""
import os

import numpy as np
import torch

from oracle import encode_ref, indel_ref, snv_ref, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_DUP = {".layer.1.": ".bn1.", ".layer.2.": ".conv1.", ".layer.4.": ".bn2.", ".layer.5.": ".conv2."}


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def expand_state(template_keys, unique):
    """Rebuild a full state dict (incl. the ResBlock duplicate keys) from the de-duplicated fixture."""
    out = {}
    for k in template_keys:
        src = k
        for a, b in _DUP.items():
            src = src.replace(a, b)
        out[k] = torch.from_numpy(np.asarray(unique[src]))
    return out


def fixture_weights(fx):
    return {k[3:]: fx[k] for k in fx.files if k.startswith("w::")}


def onehot(codes):
    return torch.from_numpy(np.ascontiguousarray(encode_ref._OHE[codes].transpose(0, 2, 1)))


def snv_oracle_from_hp(hp, drops=(0.1, 0.1, 0.25)):
    r, order, R, h1, h2, C, k, n_class = [int(v) for v in hp[:8]]
    model_no = int(hp[8]) if len(hp) > 8 else 2
    return snv_ref.build(model_no, local_radius=r, local_order=order, distal_radius=R, hidden=(h1, h2), channels=C,
                         ksize=k, n_class=n_class, emb_dropout=drops[0], local_dropout=drops[1],
                         distal_fc_dropout=drops[2])


def snv_state_for(fx, model):
    if "seed" in fx.files:
        return synth.synth_state_dict(model.state_dict(), int(fx["seed"]))
    return expand_state(model.state_dict().keys(), fixture_weights(fx))


def indel_oracle_from_hp(hp, down):
    R, C, k, n_class, rev = [int(v) for v in hp]
    return indel_ref.build(n_class=n_class, channels=C, ksize=k, down_list=[int(d) for d in down], use_reverse=bool(rev))


def indel_state_for(fx, model):
    if "seed" in fx.files:
        return synth.synth_state_dict(model.state_dict(), int(fx["seed"]))
    return {k: torch.from_numpy(v) for k, v in fixture_weights(fx).items()}
