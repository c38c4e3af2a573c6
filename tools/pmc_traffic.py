#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC passes of tools/gpu_pmc_wgrad.sh (gpurun_out/pmc_wg_{mfma,fetch,write}.txt),
run HERE after the gpurun call has merged them back.  bench.py puts `hbm_bytes` of the dominant kernel into
`roofline.traffic` only while the sha256 of the kernel sources recorded here equals that of the tree it runs
from - so the number cannot outlive the kernel it was measured on.

  hbm_bytes = 2 * FETCH_SIZE * 1024 + bytes stored      (gfx950: FETCH_SIZE counts half of a wide coalesced read,
                                                          MI355X_MICROARCH.md; KB units)
The sliding-window wgrad's WRITE_SIZE equals its FETCH_SIZE (the counter sees the LDS-DMA requests), so its stored
bytes are taken as what it does store: the per-slice slabs, nsplit * Co * Ci * Kt * 4.
"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["step_amd/csrc/tcn_wgrad_sw.hip", "step_amd/csrc/tcn_wgrad.hip", "step_amd/csrc/tcn.hip",
           "step_amd/csrc/tcn_pack.h", "step_amd/csrc/common.h", "step_amd/csrc/tcn_dft.hip"]
LINE = re.compile(r"^(.*?)\s+([A-Z_]+)\s+n=(\d+) mean=([0-9.eE+-]+) mean_dur_ns=(\d+)\s*$")
N, C, T, S, KT = 4096, 64, 240, 2, 75


def read(path):
    out = {}
    for line in open(path):
        m = LINE.match(line.rstrip("\n"))
        if m:
            name = m.group(1).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            out.setdefault(name, {})[m.group(2)] = (float(m.group(4)), int(m.group(5)))
    return out


def main(src="gpurun_out"):
    mf, fe, wr = (read(os.path.join(ROOT, src, "pmc_wg_%s.txt" % k)) for k in ("mfma", "fetch", "write"))
    h = hashlib.sha256()
    for p in SOURCES:
        h.update(open(os.path.join(ROOT, p), "rb").read())
    sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--"] + SOURCES, capture_output=True, text=True).stdout.strip()
    tout = (T - 1) // S + 1
    alg = 4 * N * C * 16 * (T + tout)
    res = {"_comment": __doc__.strip().split("\n\n")[0], "batch": N, "frames": T, "sources": SOURCES,
           "source_sha256": h.hexdigest(), "git_sha": sha + ("+dirty" if dirty else ""),
           "command": "python3 tools/kbench.py all --n 4096 --c 64 --stride 2 --t 240 --reps 1 (tools/gpu_pmc_wgrad.sh)"}
    for k in sorted(fe):
        if k not in wr or k not in mf:
            continue
        f, dur_f = fe[k]["FETCH_SIZE"]
        w, dur_w = wr[k]["WRITE_SIZE"]
        gui, dur_m = mf[k]["GRBM_GUI_ACTIVE"]
        busy = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        ent = {"kernel": k, "ms": round((dur_f + dur_w) / 2e6, 3), "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
               "mfma_busy": round(busy / 1024 / (gui / 8), 4), "clock_ghz": round(gui / 8 / dur_m, 3),
               "algorithmic_bytes": alg}
        if k.startswith("tcn_wgrad_sw_kernel"):
            # the fp32 wgrad is two launches (taps 0..59 | 60..74, tcn_wgrad_sw.hip): their bytes and times add up
            # to one "launch" of the operator; stored bytes = the per-slice slabs of both
            prev = res.get("tcn_wgrad_b1")
            stored = (42 * 60 + 128 * 16) * C * C * 4 if prev is None else 0
            ent["hbm_bytes"] = 2 * f * 1024 + stored
            if prev is not None:
                tot = prev["ms"] + ent["ms"]
                ent = {"kernel": prev["kernel"] + " + " + k, "ms": round(tot, 3),
                       "FETCH_SIZE_KB": prev["FETCH_SIZE_KB"] + ent["FETCH_SIZE_KB"],
                       "WRITE_SIZE_KB": prev["WRITE_SIZE_KB"] + ent["WRITE_SIZE_KB"],
                       "mfma_busy": round((prev["mfma_busy"] * prev["ms"] + ent["mfma_busy"] * ent["ms"]) / tot, 4),
                       "clock_ghz": round((prev["clock_ghz"] * prev["ms"] + ent["clock_ghz"] * ent["ms"]) / tot, 3),
                       "algorithmic_bytes": alg, "hbm_bytes": prev["hbm_bytes"] + ent["hbm_bytes"]}
            ent["hbm_bytes"] = float("%.4g" % ent["hbm_bytes"])
            res["tcn_wgrad_b1"] = ent
        elif k.startswith("tcn_conv_kernel"):
            ent["hbm_bytes"] = float("%.4g" % ((2 * f + w) * 1024))
            # stride 2: forward = <5, 38, 2, ...>, input gradient = two output phases of 19 taps <5, 19, 1, ...>
            res["tcn_fwd_b1" if "38, 2" in k else "tcn_dgrad_b1"] = ent
    # frequency-domain stage kernels (tools/gpu_pmc_dft.sh: one stage per run, so the per-kernel means are per launch).
    # NOTE the helper launches of a run (transform of a / dout that feed the stage) share the kernel template; the
    # stage itself is the LAST launch and the only one repeated, the means below are over all launches of that name in
    # the run and are therefore used only for the stage whose launches all have one shape.
    # block 1 runs as two stride-1 convs on the even / odd frames (dft.phases): T / 2 frames per phase on a grid of
    # rows = round_up(T / 2 + 19 + 1, 4); the input's spectrum holds both phases (2 C channels), dout's C
    rows = (T // 2 + 19 + 1 + 3) // 4 * 4
    spec_a, spec_d = 4 * N * rows * 2 * C * 16, 4 * N * rows * C * 16
    tens, tens_o = 4 * N * C * T * 16, 4 * N * C * (T // 2) * 16
    for key, st, templ, alg_d in (("dft_transform_b1", "tr", "dft_gemm_kernel<3, false, 2>", tens + spec_a),
                                  ("dft_inverse_bwd_b1", "ib", "dft_gemm_kernel<2, true, 2>", spec_a + tens),
                                  ("dft_mix_b1", "mf", "dft_mix_kernel<2, 128>", spec_a + spec_d),
                                  ("dft_mix_bwd_b1", "mb", "dft_mix_kernel<4, 64>", spec_d + spec_a),
                                  ("dft_wgrad_b1", "wg", "dft_wgrad_lds_kernel<2, 4>", spec_d + spec_a)):
        try:
            mfd, fed, wrd = (read(os.path.join(ROOT, src, "pmc_dft_%s_%s.txt" % (st, k))) for k in ("mfma", "fetch", "write"))
        except OSError:
            continue
        if templ not in fed or templ not in wrd or templ not in mfd:
            continue
        f, dur_f = fed[templ]["FETCH_SIZE"]
        w, dur_w = wrd[templ]["WRITE_SIZE"]
        gui, dur_m = mfd[templ]["GRBM_GUI_ACTIVE"]
        busy = mfd[templ]["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        res[key] = {"kernel": templ, "ms": round((dur_f + dur_w) / 2e6, 3), "FETCH_SIZE_KB": round(f, 1),
                    "WRITE_SIZE_KB": round(w, 1), "mfma_busy": round(busy / 1024 / (gui / 8), 4),
                    "clock_ghz": round(gui / 8 / dur_m, 3), "algorithmic_bytes": alg_d,
                    "hbm_bytes": float("%.4g" % ((2 * f + w) * 1024)),
                    "command": "python3 tools/dft_bench.py --reps 1 --no-direct --only ... (tools/gpu_pmc_dft.sh)"}
    json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for k in ("dft_transform_b1", "dft_inverse_bwd_b1", "dft_mix_b1", "dft_mix_bwd_b1", "dft_wgrad_b1"):
        if k in res:
            e = res[k]
            print("%-18s %-30s %7.3f ms  hbm %.3g B (algorithmic %.3g)  mfma busy %.3f @ %.2f GHz" %
                  (k, e["kernel"], e["ms"], e["hbm_bytes"], e["algorithmic_bytes"], e["mfma_busy"], e["clock_ghz"]))
    for k in ("tcn_wgrad_b1", "tcn_fwd_b1", "tcn_dgrad_b1"):
        if k in res:
            e = res[k]
            print("%-14s %-40s %7.3f ms  hbm %.3g B (algorithmic %.3g)  mfma busy %.3f @ %.2f GHz" %
                  (k, e["kernel"], e["ms"], e["hbm_bytes"], e["algorithmic_bytes"], e["mfma_busy"], e["clock_ghz"]))


if __name__ == "__main__":
    main(*sys.argv[1:])
