#!/usr/bin/env python3
"""BASELINE configs[0] through the HIP modules: train -> save -> reload -> evaluate the real_only ST-GCN classifier
the way classifier_stgcn_real_only/main.py + utils/processor.py do (342 gaits at T = 75, batch 8, drop_last,
Adam(lr 0.1, wd 1e-4), lr x0.1 at ceil(E * [0.5, 0.75, 0.875]), top-1 after every epoch, a checkpoint
`epoch{E}_acc{A:.2f}_model.pth.tar` on every improvement, best one reloaded by parsing the file name).

The real Emotion-Gait data is an external download (README.md:16) that is not in this image: the gaits are synthetic
(step_amd.harness.synthetic_gaits).  Needs a GPU (there is no CPU path in step_amd).

    python tools/train_classifier.py --num-epoch 8 --work-dir /tmp/step_run
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-epoch", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--base-lr", type=float, default=0.1)
    ap.add_argument("--weight-decay", type=float, default=1e-4)
    ap.add_argument("--gaits", type=int, default=342)
    ap.add_argument("--frames", type=int, default=75)
    ap.add_argument("--work-dir", default="/tmp/step_amd_train")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variant", choices=["real_only", "real_and_synth"], default="real_only")
    args = ap.parse_args()

    from step_amd import harness
    from step_amd.net import classifier, classifier_real_and_synth
    torch.manual_seed(args.seed)
    mod = classifier if args.variant == "real_only" else classifier_real_and_synth
    model = mod.Classifier(3, 4, {"strategy": "spatial"})
    train_set, test_set = harness.synthetic_gaits(args.gaits, args.frames, 4, args.seed)
    run = harness.EpochRunner(model, train_set, test_set, args.work_dir, args.num_epoch, args.batch_size,
                              args.base_lr, args.weight_decay, seed=args.seed, log=lambda s: print(s, flush=True))
    t0 = time.perf_counter()
    hist = run.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    path = run.load_best_model()
    loss, acc = run.per_test()
    print(json.dumps({"epochs": len(hist), "seconds": round(dt, 2), "iterations": run.meta["iter"],
                      "sequences_per_s": round(run.meta["iter"] * args.batch_size / dt, 1),
                      "best_checkpoint": os.path.basename(path), "reloaded_top1": acc[1], "reloaded_loss": loss,
                      "first_train_loss": hist[0]["train_loss"], "last_train_loss": hist[-1]["train_loss"]}))


if __name__ == "__main__":
    main()
