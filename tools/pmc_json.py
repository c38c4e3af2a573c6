#!/usr/bin/env python3
"""Merge the text summaries written by tools/gpu_pmc_gcn.sh (gpurun_out/pmc_hbm{1,2}_{mfma,fetch,write}.txt) into
one JSON (profiles/r1_pmc_hbm_kernels_final.json).  Only kernels of this library with >= 0.1 ms are kept.

  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024     (gfx950 correction, MI355X_MICROARCH.md)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
"""
import json
import re
import sys

LINE = re.compile(r"^(.*?)\s+([A-Z_]+)\s+n=(\d+) mean=([0-9.eE+-]+) mean_dur_ns=(\d+)\s*$")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].strip()


def read(path):
    out = {}
    for line in open(path):
        m = LINE.match(line.rstrip("\n"))
        if m:
            out.setdefault(short(m.group(1)), {})[m.group(2)] = (float(m.group(4)), int(m.group(5)))
    return out


def main(src="gpurun_out", dst="profiles/r1_pmc_hbm_kernels_final.json", comment=""):
    res = {"_comment": comment}
    for b in (1, 2):
        mf = read(f"{src}/pmc_hbm{b}_mfma.txt")
        fe = read(f"{src}/pmc_hbm{b}_fetch.txt")
        wr = read(f"{src}/pmc_hbm{b}_write.txt")
        for k in sorted(fe):
            if k not in wr or k not in mf or not (k.startswith(("gcn_", "bn_", "affine_")) and fe[k]["FETCH_SIZE"][1] >= 100000):
                continue
            f, dur_f = fe[k]["FETCH_SIZE"]
            w, dur_w = wr[k]["WRITE_SIZE"]
            gui, dur_m = mf[k]["GRBM_GUI_ACTIVE"]
            busy = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0]
            ns = (dur_f + dur_w) / 2
            hbm = (2 * f + w) * 1024
            res[f"block{b}:{k}"] = {
                "ms": round(ns / 1e6, 3),
                "mfma_busy": round(busy / 1024 / (gui / 8), 4),
                "clock_ghz": round(gui / 8 / dur_m, 3),
                "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                "hbm_bytes": float(f"{hbm:.4g}"),
                "hbm_gbs": round(hbm / ns, 1),
                "frac_hbm": round(hbm / ns / 8000, 3),
            }
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        if k != "_comment":
            print(f"{k:48s} {v['ms']:7.3f} ms  hbm {v['frac_hbm']:.3f}  mfma {v['mfma_busy']:.3f}")


if __name__ == "__main__":
    old = json.load(open("profiles/r1_pmc_hbm_kernels_final.json")).get("_comment", "")
    main(comment=old if len(sys.argv) < 2 else sys.argv[1])
