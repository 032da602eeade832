"""End-to-end composition on the GPU: FASTA + BED -> training batches -> epochs (the reference's loop policy) -> validation
predictions -> full-Dirichlet fit, calibration metrics, k-mer / regional analytics -> checkpoint files -> reload -> file-level
prediction table.  Every piece has its own parity test; this one checks that they fit together the way
MuRaL/training.py:380-520 and scripts/run_predict.py:188-239 chain them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_evaluate_save_reload_predict(tmp_path):
    from mural_amd import evaluation as E
    from mural_amd import train as TR
    from mural_amd.calibration import load_dirichlet_weights
    from mural_amd.data import ingest
    from mural_amd.model import model_choice, nn_utils, weights_init
    rng = np.random.default_rng(2024)
    r, R, n_class = 4, 110, 4
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=60000).tobytes().decode()
    arr = np.frombuffer(seq.encode(), np.uint8)
    fa = tmp_path / "g.fa"
    fa.write_text(">chr1\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + "\n")
    # A sites on '+', T on '-'; the mutation rate depends on the downstream base, so there is something to learn
    cand = np.sort(rng.choice(np.arange(R + 1, len(seq) - R - 1), size=9000, replace=False))
    rows = []
    for p in cand:
        if arr[p] not in (ord("A"), ord("T")):
            continue
        st = "+" if arr[p] == ord("A") else "-"
        nxt = arr[p + 1] if st == "+" else {65: 84, 67: 71, 71: 67, 84: 65}[int(arr[p - 1])]
        rate = 0.45 if nxt == ord("G") else 0.08
        lab = int(rng.integers(1, n_class)) if rng.random() < rate else 0
        rows.append((int(p), st, lab))
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"chr1\t{p}\t{p + 1}\t.\t{lab}\t{st}\n" for p, st, lab in rows))

    ncol = 2 * r + 1 - 2
    config = dict(local_radius=r, local_order=3, local_hidden1_size=150, local_hidden2_size=75, distal_radius=R, emb_dropout=0.1,
                  local_dropout=0.1, CNN_kernel_size=3, CNN_out_channels=32, distal_fc_dropout=0.25, n_class=n_class, model_no=2,
                  seq_only=True, emb_dims=[(65, 2)] * ncol, segment_center=5000, optim="Adam", learning_rate=2e-3, weight_decay=1e-6,
                  lr_scheduler="StepLR", LR_gamma=0.9, batch_size=256, min_lr=1e-6, restart_lr=1e-4)
    common = dict(emb_dims=config["emb_dims"], n_cont=0, n_class=n_class, distal_order=1, in_channels=4)
    torch.manual_seed(0)
    model = model_choice(2, config, common, "snv")
    model.apply(weights_init)
    model = model.cuda()
    opt = TR.make_optimizer(config, model.parameters())
    sch = TR.make_scheduler(config, opt)
    crit = torch.nn.CrossEntropyLoss(reduction="sum")
    losses = []
    for epoch in range(3):
        batches = ingest.train_batches_from_files(fa, bed, config["batch_size"], r, 3, R, segment_center=config["segment_center"],
                                                  sampled_segments=4, shuffle=True, generator=torch.Generator().manual_seed(epoch))
        losses.append(TR.train_epoch(model, batches, crit, opt, sch, config, "cuda", epoch=epoch) / len(rows))
    assert losses[-1] < losses[0], losses

    # validation-style pass on the same sites: predictions, calibrator, analytics
    model.eval()
    res = ingest.predict_bed(model, fa, bed, r, 3, segment_center=config["segment_center"])
    prob = torch.from_numpy(res["prob"]).cuda()
    label = torch.from_numpy(res["label"].astype(np.int64)).cuda()
    assert np.allclose(res["prob"].sum(axis=1), 1.0, atol=1e-5)
    weights, nll_cal, prob_cal = E.calibrate_prob(prob, label, printer=lambda *a: None)
    assert nll_cal <= E.calibration_metrics(prob, label)["nll"] + 1e-9         # the fit cannot be worse than the identity map
    genome = ingest.read_fasta(fa, "cuda")["chr1"]
    pos = torch.from_numpy(res["start"]).cuda()
    strand = torch.from_numpy((res["strand"] == "-").astype(np.uint8)).cuda()
    codes = genome.encode_kmer(pos, strand, r, 1)
    ev = E.Evaluator(codes, label, prob_cal, n_class, calibra="FullDiri", printer=lambda *a: None)
    kmer = ev.evaluate_kmer([3, 5])
    assert kmer[3][0] > 0.5                      # the model picked up the planted dinucleotide effect
    corr = ev.evaluate_regional_corr(torch.zeros_like(pos, dtype=torch.int32), pos, win_size_list=(5000,))
    assert len(corr[5000]) == n_class

    # checkpoint files of training.py:570-578, reload, file-level prediction table
    ckpt = str(tmp_path / "model")
    nn_utils.save_model(model, weights, config, ckpt)
    again, cfg = nn_utils.load_model(ckpt)
    res2 = ingest.predict_bed(again, fa, bed, cfg["local_radius"], cfg["local_order"], segment_center=cfg["segment_center"])
    assert np.array_equal(res2["prob"], res["prob"])
    assert ingest.write_predictions(res2, tmp_path / "pred.tsv", dirichlet_weights=load_dirichlet_weights(ckpt + ".fdiri_cal.pkl")) == len(rows)
    import pandas as pd
    table = pd.read_csv(tmp_path / "pred.tsv", sep="\t")
    assert list(table.columns[:5]) == ["chrom", "start", "end", "strand", "mut_type"] and len(table) == len(rows)
    assert table["start"].is_monotonic_increasing
    # the same through the command-line helper
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("predict_files", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                  "tools", "predict_files.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main([ckpt, str(fa), str(bed), str(tmp_path / "pred2.tsv")])
    assert (tmp_path / "pred2.tsv").read_text() == (tmp_path / "pred.tsv").read_text()


def test_indel_train_from_files_and_predict(tmp_path):
    """The INDEL model through the same chain: training batches from FASTA + BED (indel windows), two optimiser steps per
    epoch policy, Poisson-calibrated prediction table (run_predict.py:224-225 applies it to every INDEL model)."""
    from mural_amd import train as TR
    from mural_amd.data import ingest
    from mural_amd.model import model_choice, weights_init
    rng = np.random.default_rng(7)
    R, n_class = 1000, 3
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=12000).tobytes().decode()
    fa = tmp_path / "g.fa"
    fa.write_text(">chr1\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + "\n")
    pos = np.sort(rng.choice(np.arange(10, len(seq) - 10), size=48, replace=False))
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"chr1\t{p}\t{p + 1}\t.\t{int(rng.integers(0, n_class))}\t{'+-'[int(rng.integers(0, 2))]}\n" for p in pos))
    config = dict(local_radius=4, local_order=3, distal_radius=R, CNN_kernel_size=7, CNN_out_channels=8, down_list=[1, 4, 5, 5, 5, 2],
                  use_reverse=True, n_class=n_class, model_no=0, optim="AdamW", learning_rate=1e-3, weight_decay=1e-6,
                  lr_scheduler="StepLR", LR_gamma=0.9, batch_size=16, min_lr=1e-7, restart_lr=1e-4, segment_center=4000)
    torch.manual_seed(1)
    model = model_choice(0, config, dict(n_class=n_class), "indel")
    model.apply(weights_init)
    model = model.cuda()
    opt = TR.make_optimizer(config, model.parameters())
    sch = TR.make_scheduler(config, opt)
    crit = torch.nn.CrossEntropyLoss(reduction="sum")
    losses = []
    for epoch in range(4):
        batches = ingest.train_batches_from_files(fa, bed, 16, 4, 3, R, segment_center=4000, sampled_segments=2, shuffle=True,
                                                  generator=torch.Generator().manual_seed(epoch), model_type="indel")
        losses.append(TR.train_epoch(model, batches, crit, opt, sch, config, "cuda", model_type="indel", epoch=epoch))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    model.eval()
    res = ingest.predict_bed(model, fa, bed, 4, 3, distal_radius=R, segment_center=4000, model_type="indel")
    assert res["prob"].shape == (48, n_class) and np.allclose(res["prob"].sum(axis=1), 1.0, atol=1e-5)
    import pandas as pd
    assert ingest.write_predictions(res, tmp_path / "indel.tsv", poisson=True) == 48
    table = pd.read_csv(tmp_path / "indel.tsv", sep="\t")
    assert len(table) == 48 and np.isfinite(table[["prob0", "prob1", "prob2"]].to_numpy()).all()
