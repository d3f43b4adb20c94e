#!/usr/bin/env python3
"""Rewrites the measurement table of DESIGN.md section 5 (between the two marker comments) from profiles/r6_bench_n1.json (development aid)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.loads(open(os.path.join(ROOT, "profiles", "r6_bench_n1.json")).read().strip().splitlines()[-1])
a, r, api, e2e, cpu = d["also"], d["roofline"], d["api_end_to_end"], d["end_to_end"], d["cpu_baseline"]
rows = ["| | ms per step | Mvox/s | dominant kernel, `frac` | notes |", "|---|---|---|---|---|"]
rows.append("| headline: 512³, 6-conn, diff_exp σ = 15 | **%.2f** (17.7 – 18.2 over the boxes; round 5: 18.61 on the driver's box, round 4: 35.75) | **%d** | `k_discharge_w` %.1f µs × 58 (HIP events; rocprofv3 kernel trace of the same command: `r6_kernel_stats.csv`), **%.3f**; `traffic` %.0f MB per launch = %.2f × the algorithmic bytes | build %.2f, discharges %.2f, relabels %.2f ms; 4 global relabels, 58 colour phases, 537 k tile discharges; labels = SHA-256 of the reference's label volume (`labels_match_reference`), validation block all zero |" % (
    d["ms_per_step"], round(d["value"]), r["avg_launch_ms"] * 1e3, r["frac"], r["traffic"] / 1e6, r["traffic"] / (r["voxels_per_launch"] * 71.0),
    d["phases_ms"]["build"], d["phases_ms"]["discharge_kernels"], d["phases_ms"]["relabel_kernels"]))


def row(name, key, note):
    v = a[key]
    rows.append("| `also.%s`: %s | %.2f | %d | %.3f | %s |" % (key, name, v["ms_per_step"], round(v["mvoxels_s"]), v["frac"], note))


row("256³", "config2", "labels = the reference's (`sphere_256_6`); launch-bound (a launch is one visit deep)")
row("512³, 26-conn + regional map", "config3", "`k26_discharge_w`; round 5: 44.9 (the relabel with a regional term, §6b; the cut value in two launches; every arc pair's weight evaluated once in `k_build<true>`); `config3_at_256` in the same line checks the reference's label hash at 256³; counter traffic 553 MB per launch = 2.2 × (`pmc_discharge26.json`)")
row("weak contrast 512³", "hard", "no walls → exact labels, repeated in-plane steps (round 5: 64.4)")
row("tie-heavy 512³ (markers everywhere)", "ties", "round 5: 697")
row("26-conn, markers only, 512³", "conn26_markers", "round 5: 236")
row("uint16 CT-like 512³ (term by table)", "ct_uint16", "before the repeated in-plane steps: 74")
rows.append("| `api_end_to_end`: `graph_from_voxels(...).maxflow(); .labels()` from host arrays | %.1f per volume (%.0f – %.0f over the five calls; round 5 on the driver's box: 57 – 845) | %d | — | slowest call: `graph_from_voxels` %.1f (805 MB up), `maxflow` %.1f, `labels` %.1f; handles take their memory from the library's pool.  `end_to_end` (resident handle: upload + step + download): %.1f + %.1f + %.1f = %.1f ms |" % (
    api["ms_per_volume"], min(api["each_ms"]), max(api["each_ms"]), round(api["mvoxels_s"]), api["slowest_call_ms"]["graph_from_voxels (handle + H2D + build)"],
    api["slowest_call_ms"]["maxflow"], api["slowest_call_ms"]["labels (read-out + D2H)"], e2e["h2d_ms"], d["ms_per_step"], e2e["d2h_ms"], e2e["ms_per_volume"]))
rows.append("| `cpu_baseline` | %.1f s | %.2f | — | the reference's BK on the WHOLE 512³ in the same run, one core (NumPy weights + `sum_edge` build %.1f s + max-flow %.1f s): the GPU path is %d × |" % (
    cpu["t_build_s"] + cpu["t_solve_s"], cpu["value"], cpu["t_build_s"], cpu["t_solve_s"], round(d["value"] / cpu["value"])))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"<!-- measurement table -->.*?<!-- /measurement table -->", "<!-- measurement table -->\n" + "\n".join(rows) + "\n<!-- /measurement table -->", s, flags=re.S)
sl = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r6_slab_scaling_one_gpu.jsonl")) if l.startswith("{")]
srows = ["| slabs | Σ kernel ms per slab | relabel passes | phases | exchanges | reductions |", "|---|---|---|---|---|---|"]
for x in sl:
    k = x["kernel_ms_per_slab"]
    srows.append("| %d%s | %s | %s | %d | %s | %s |" % (x["slabs"], " (`mgc_maxflow`)" if x["slabs"] == 1 else "", ("%.0f" % k[0]) if len(k) == 1 else "%.0f – %.0f" % (min(k), max(k)),
                                                       x.get("relabel_passes", x.get("relabel_launches", "–")), x["phases"], x.get("exchanges", "–"), x.get("reductions", "–")))
s = re.sub(r"<!-- slab table -->.*?<!-- /slab table -->", "<!-- slab table -->\n" + "\n".join(srows) + "\n<!-- /slab table -->", s, flags=re.S)
open(p, "w").write(s)
print("\n".join(rows)[:600])
print("\n".join(srows))
