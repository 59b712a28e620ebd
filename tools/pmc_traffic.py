#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the counter summary tools/pmc.sh wrote (gpurun_out/pmc_summary.txt)"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_summary.txt")
vals = {}
for ln in open(src):
    m = re.match(r"\s+(\S.*?)\s{2,}(\S.*)$", ln)
    if not m: continue
    name = m.group(1).strip()
    for kv in m.group(2).split():
        if "=" in kv:
            k, v = kv.split("=")
            vals.setdefault(name, {})[k] = float(v)
# (kernel names are matched by prefix: template argument lists grow)
keys = {"lorenzo_quant_hist": "k_lorenzo_quant_march3q<4>",  # (round 5: the 16-bit form; round 3-4: "k_lorenzo_quant_march3<float, 3, 4>")
        "chunk_bits": "k_chunk_bits2", "seg_chunks": "k_seg_chunks", "scan_groups": "k_scan_groups", "pack": "k_pack",
        "decode_with_x_scan": "k_decode<4, true", "scan_strided": "k_scan_strided_half<false, false>", "scan_strided_dequant": "k_scan_strided_half<true, false>"}
out = {}
for k, name in keys.items():
    v = next((vals[n] for n in sorted(vals) if n.startswith(name)), None)
    if not v: continue
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out[k + "_FETCH_SIZE_KB_raw"] = f
    out[k + "_WRITE_SIZE_KB_raw"] = w
    out[k + "_hbm_bytes_per_launch"] = int((2 * f + w) * 1024)
    out[k + "_valu_wave_instructions"] = v.get("SQ_INSTS_VALU", 0.0)
    out[k + "_valu_issue_us"] = round(v.get("SQ_INSTS_VALU", 0.0) * 4 / 1024 / 2.1e3, 1)
# the warm C2 step (round 3): stage 1 (probe inside), segments -> chunk counts, offset scan (+ histogram fold), packer (+ code book, list sorts)
step = [out.get(k + "_hbm_bytes_per_launch", 0) for k in ("lorenzo_quant_hist", "seg_chunks", "scan_groups", "pack")]
if all(step[:1]) and step[3]:
    out["c2_step_hbm_bytes"] = int(sum(step))
    out["c2_step_note"] = ("warm step = k_lorenzo_quant_march3 + k_seg_chunks + k_scan_groups + k_pack (the bits pass k_chunk_bits2 and the code book's own "
                           "launch only run on a first call or after a miss); algorithmic bytes of the step = input + payload = 605.8 MB")
# C3 (tools/pmc_c3.sh -> gpurun_out/pmc_summary_c3.txt): HBM bytes of stage 1 per step = all interpolation kernels + the working copy
c3 = os.path.join(os.path.dirname(src), os.path.basename(src).replace("pmc_summary", "pmc_summary_c3"))
if os.path.exists(c3):
    v3 = {}
    calls = {}
    for ln in open(c3):
        m = re.match(r"\s+(\S.*?)\s{2,}(\S.*)$", ln)
        if not m: continue
        for kv in m.group(2).split():
            if "=" in kv:
                k, v = kv.split("=")
                v3.setdefault(m.group(1).strip(), {})[k] = float(v)
    per = {n: int((2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024) for n, d in v3.items()
           if n.startswith(("k_interp", "k_hist_codes", "k_pack", "k_chunk_bits", "k_decode"))}
    out["c3_per_kernel_hbm_bytes_per_launch"] = per
    # stage 1 per step: the compression-side interpolation kernels + the code histogram; launches per step from the kernel
    # statistics of the same bench command (k_hist_codes runs once per compression)
    import csv
    stats = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r05c_kernel_stats_c3.csv", "r05b_kernel_stats_c3.csv", "r05_kernel_stats_c3.csv", "r03_kernel_stats_c3.csv", "r02_kernel_stats_c3.csv"))
                  if os.path.exists(q)), "")
    if os.path.exists(stats):
        calls = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(stats))}
        def ncalls(prefix):
            return sum(c for n, c in calls.items() if n.replace("void ", "").startswith(prefix))
        per_step = ncalls("k_hist_codes")
        detail, total = {}, 0
        for n, b in per.items():
            if not n.startswith(("k_interp_anchors", "k_interp_pass<float, false>", "k_interp_level<float, false>", "k_interp_vec<float, false", "k_hist_codes")):
                continue
            lps = ncalls(n.split("(")[0]) / max(per_step, 1)
            detail[n] = {"launches_per_step": round(lps, 2), "bytes_per_launch": b}
            total += lps * b
        out["c3_stage1_hbm_bytes_per_step"] = int(total)
        out["c3_stage1_detail"] = detail
        out["c3_stage1_note"] = ("stage 1 of C3 per step = interpolation kernels (launches per step from profiles/r05b_kernel_stats_c3.csv (the newest there is) x mean bytes per "
                                 "launch) + the code histogram; the level kernels read the input in place (no working copy)")
# C4 slab with its specified predictor set (tools/pmc_blk.sh -> profiles/r03_pmc_blk_select.txt, the benchmark field: handed to the plain path)
blk = os.path.join(ROOT, "profiles", "r03_pmc_blk_select.txt")
if os.path.exists(blk):
    import ast
    per = {}
    for ln in open(blk):
        m = re.match(r"(k_\w+) (\{.*\})\s*$", ln.strip())
        if not m: continue
        for k, v in ast.literal_eval(m.group(2)).items():
            per.setdefault(m.group(1), {})[k] = float(v)
    tot = 0
    for name in ("k_blk_select", "k_lorenzo_quant_march"):
        if name in per:
            # the doubling is calibrated for 16 B per lane only: the selection pass loads 8 B per lane
            # (a block per lane), and doubled its count would mean 5.9 TB/s over a 0.42 ms kernel
            wide = 2 if name == "k_lorenzo_quant_march" else 1
            b = int((wide * per[name].get("FETCH_SIZE", 0) + per[name].get("WRITE_SIZE", 0)) * 1024)
            out["c4_composed_%s_hbm_bytes_per_launch" % name] = b
            tot += b
    if tot:
        out["c4_composed_stage1_hbm_bytes_per_step"] = tot
        out["c4_composed_note"] = ("C4 slab (128x1024x1024 f64, 1e-6), Lorenzo + regression, benchmark field: stage 1 = selection pass + plain Lorenzo kernel; "
                                   "algorithmic bytes 2 x 8 + 2 per element = 2.42 GB; FETCH_SIZE doubled for the Lorenzo kernel (16 B per lane), taken as reported for the "
                                   "selection pass (8 B per lane: uncalibrated width, and doubled it would exceed what the kernel's 0.42 ms can carry)")
blk4a = os.path.join(ROOT, "profiles", "r03_pmc_blk_c4a.txt")
if os.path.exists(blk4a):
    import ast
    per = {}
    for ln in open(blk4a):
        m = re.match(r"(k_\w+) (\{.*\})\s*$", ln.strip())
        if not m: continue
        for k, v in ast.literal_eval(m.group(2)).items():
            per.setdefault(m.group(1), {})[k] = float(v)
    tot = 0
    for name in ("k_blk_select", "k_blk_fit", "k_blk_rows"):
        if name in per and "FETCH_SIZE" in per[name]:
            # (8-byte-per-lane loads in all three: the counter taken as reported, like the selection pass above)
            b = int((per[name].get("FETCH_SIZE", 0) + per[name].get("WRITE_SIZE", 0)) * 1024)
            out["c4a_composed_%s_hbm_bytes_per_launch" % name] = b
            tot += b
    if tot:
        out["c4a_composed_stage1_hbm_bytes_per_step"] = tot
        out["c4a_composed_note"] = ("C4 slab, C4a field (regression in 14 % of the blocks): stage 1 = selection pass + regression blocks (k_blk_fit) + "
                                    "Lorenzo elements by rows of blocks (k_blk_rows); the side section's kernels move a few MB; algorithmic bytes 2.42 GB")
out["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 20 --warmup 3` (tools/pmc.sh); "
                 "mean per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide (16 B/lane) coalesced reads on gfx950; "
                 "WRITE_SIZE taken as reported (checks out: stage 1 writes 1 B/elem of codes = 134 MB, counter says 135 MB); counters are in KB (x1024). "
                 "For k_decode (4-byte loads scattered over 64 lines per wave) the doubling over-counts. valu_issue_us = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.1 GHz: "
                 "the time the vector ALUs need to issue the kernel's instructions (a wave64 instruction occupies a 16-lane SIMD for 4 cycles). "
                 "Generated by tools/pmc_traffic.py from profiles/r05b_pmc_summary.txt (C2, round 5) and profiles/r05c_pmc_summary_c3.txt (C3, round 5 with the dense hand-over between the two finest levels: tools/pmc_c3.sh; launches per step from profiles/r05c_kernel_stats_c3.csv).")
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:600])
