"""Deterministic synthetic parameters for parity tests and the bench -- TEST INFRASTRUCTURE ONLY.

Fills any ``state_dict`` (oracle, product or reference model -- they share keys) from a numpy PCG64
stream keyed on (seed, key order), so that fixtures need to store only *outputs*: the weights are
regenerated bit-identically wherever the test runs.  Scales mimic a trained network (non-trivial BN
statistics, xavier/kaiming-like weight magnitudes) so every folded term is exercised.
"""
import numpy as np
import torch


def synth_state_dict(template: dict, seed: int) -> dict:
    rng = np.random.default_rng(seed)
    out = {}
    done = {}
    for key, ref in template.items():
        shape = tuple(ref.shape)
        # the SNV ResBlock registers its modules twice (bn1 == layer.1 ...): reuse the first draw
        alias = (key.replace(".layer.1.", ".bn1.").replace(".layer.2.", ".conv1.")
                    .replace(".layer.4.", ".bn2.").replace(".layer.5.", ".conv2."))
        if alias != key and alias in done:
            out[key] = done[alias]
            continue
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.tensor(7, dtype=torch.int64)
        elif leaf == "running_mean":
            t = torch.from_numpy(rng.normal(0.0, 0.2, shape).astype(np.float32))
        elif leaf == "running_var":
            t = torch.from_numpy(rng.uniform(0.4, 1.6, shape).astype(np.float32))
        elif leaf == "weight" and len(shape) == 1:      # BN gamma
            t = torch.from_numpy(rng.uniform(0.6, 1.4, shape).astype(np.float32))
        elif leaf == "bias":
            t = torch.from_numpy(rng.normal(0.0, 0.1, shape).astype(np.float32))
        elif leaf == "weight" and key.startswith(("emb_layer", "model.emb_layer")):
            t = torch.from_numpy(rng.normal(0.0, 1.0, shape).astype(np.float32))
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = torch.from_numpy(rng.normal(0.0, np.sqrt(1.6 / max(fan_in, 1)), shape).astype(np.float32))
        else:
            raise KeyError(f"unhandled state key {key}")
        out[key] = t
        done[key] = t
    return out
