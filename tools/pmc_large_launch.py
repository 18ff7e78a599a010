"""HBM traffic, L2 hit rate, clock and MFMA-busy share of the configs[4] in_proj gradient group from its PMC summary
(tools/pmc_summary.py output of the separate --pmc passes; tools/gpu/profile.sh section `big`):
    python tools/pmc_large_launch.py <tag>_pmc_plane_gemm_cfg5_raw.txt <tag>_pmc_large_launch_traffic.json [us per launch]
Units per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 (gfx950 counts 128-B requests as 64 B on wide
streaming reads), WRITE_SIZE exact; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import json, re, sys
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"\s+(\w+)\s+avg\s+([0-9.]+)", line)
    if m:       # (the pair may be TWO launches -- a (kernel, grid) entry each per counter pass: their per-launch averages add up to the pair's)
        vals[m.group(1)] = vals.get(m.group(1), 0.0) + float(m.group(2))
M, Nout, Kin = 16384, 3072, 1024
fetch, write = vals.get("FETCH_SIZE", 0.0) * 1024 * 2, vals.get("WRITE_SIZE", 0.0) * 1024
alg = 4.0 * (M * Nout + M * Kin + Nout * Kin) + 4.0 * (M * Kin + Nout * Kin) + 4.0 * M * Kin
out = {"kernel": "gemm_planes_kernel<3> dgrad + wgrad of one dY [16384x3072]x[3072x1024] (configs[4] in_proj) as the plans launch it -- slnlp_gemm_wd: a launch each, "
                 "256x256 tiles, wgrad split-K 5 with its slices added by plane_splitk_reduce_kernel (tools/bench_plane_one.py 16384 3072 1024 0 0 3); counters summed over the pair's THREE launches",
       "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write, "algorithmic_bytes": alg, "ratio": round((fetch + write) / alg, 2),
       "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE exact; KiB -> bytes; separate --pmc passes"}
if "TCC_HIT_sum" in vals:
    out["l2"] = {"TCC_HIT_sum": vals["TCC_HIT_sum"], "TCC_MISS_sum": vals["TCC_MISS_sum"],
                 "hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 3)}
if "GRBM_GUI_ACTIVE" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    out["cycles_per_launch"] = cyc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        out["mfma_busy_of_measured_cycles"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 3)
    if len(sys.argv) > 3:
        out["us_per_launch"] = float(sys.argv[3])
        out["clock_ghz_from_grbm"] = round(cyc / float(sys.argv[3]) / 1e3, 3)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
