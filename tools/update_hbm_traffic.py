#!/usr/bin/env python3
"""profiles/hbm_traffic.json <- the L2 / HBM counters of THIS round's run of the dominant kernels.

    python tools/update_hbm_traffic.py <tcc.csv of g_a.2> [<tcc.csv of g_a.0 + GDN>] --tag r06

tcc.csv = tools/pmc_summary.py over tools/debug/prof_tcc.sh's separate `rocprofv3 --pmc` passes (TCC_HIT_sum TCC_MISS_sum |
FETCH_SIZE | WRITE_SIZE | ...).  HBM bytes per launch = FETCH_SIZE [KB] x 1024 x 2 + WRITE_SIZE [KB] x 1024: on gfx950 FETCH_SIZE
counts 64 B per 128-byte request of the kernels' 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is
exact.  The entry names the kernel as the profiler saw it (template arguments included), the file it came from and the round;
bench.py's `roofline.traffic` / `traffic_source` read these keys."""
import argparse
import csv
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def row_of(path, needle):
    rows = [r for r in csv.DictReader(open(path)) if needle in r["kernel"]]
    if not rows:
        raise SystemExit(f"{path}: no kernel matching {needle!r}")
    return max(rows, key=lambda r: float(r.get("FETCH_SIZE") or 0))


def entry(path, needle, what, algorithmic, note, tag):
    r = row_of(path, needle)
    fetch, write = float(r["FETCH_SIZE"]), float(r["WRITE_SIZE"])
    hit, miss = float(r.get("TCC_HIT_sum") or 0), float(r.get("TCC_MISS_sum") or 0)
    b = fetch * 1024 * 2 + write * 1024
    return {"kernel": f"{r['kernel']} (grid {r['grid']}, workgroup {r['workgroup']}) = {what}", "round": tag, "source": os.path.relpath(path, REPO),
            "launches_averaged": int(r["launches"]), "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
            "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "bytes_per_launch": b, "algorithmic_bytes_per_launch": algorithmic,
            "traffic_over_algorithmic": b / algorithmic, "algorithmic_note": note}, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ga2")
    ap.add_argument("c4gdn", nargs="?")
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    path = os.path.join(REPO, "profiles", "hbm_traffic.json")
    d = json.load(open(path))
    e, b = entry(a.ga2, "conv_f16x3_kernel", "g_a.2 + fused GDN g_a.3 (B=16, 192->192, 5x5 s2, 128^2->64^2), two fp16 planes in -> planes out",
                 255492096, "input planes 16x128x128x192 x 4 B + packed weights 150 chunks x 24 KiB + gamma + output planes 16x64x64x192 x 4 B", a.tag)
    d["g_a2_f16x3"], d["g_a2_f16x3_bytes_per_launch"] = e, b
    if a.c4gdn:
        e, b = entry(a.c4gdn, "c4gdn_f16x3_kernel", "g_a.0 + GDN g_a.1 (B=16, 3->192, 5x5 s2, 256^2->128^2), NHWC4 image in -> two fp16 planes out",
                     218349568, "NHWC4 image 16.8 MB + output planes 201 MB + A-operand stream 0.25 MB", a.tag)
        d["g_a0_c4gdn_f16x3"], d["g_a0_c4gdn_bytes_per_launch"] = e, b
    json.dump(d, open(path, "w"), indent=1)
    print(json.dumps({k: d[k] for k in ("g_a2_f16x3", "g_a2_f16x3_bytes_per_launch")}, indent=1))


if __name__ == "__main__":
    main()
