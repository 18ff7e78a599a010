"""Derived MFMA utilisation of the `roofline` kernel (the grouped dgrad + wgrad plane GEMM) from rocprofv3 --pmc passes of
tools/bench_group.py, reconciled with its FLOP rate.

  issued MFMA instructions  = SQ_INSTS_VALU_MFMA_MOPS_BF16 (counted in units of 512 FLOP... see below) or, analytically,
                              tile-K-steps x 8 waves x 12 v_mfma_f32_16x16x32_bf16 (gemm_planes.hip: 2 kk x (3 passes x 2 n-frags))
  cycles one of them holds its SIMD's matrix pipe = 16 (MI355X_MICROARCH.md, cycle constants: 16x16x32 bf16 back to back)
  available = 1024 SIMDs x kernel duration x shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration)
  utilisation (issued)  = issued x 16 / available          utilisation (useful) = issued-utilisation / 3 passes

usage: mfma_util.py <pmc_dir> <workgroups> <M_tokens> <N_out> <K_in> <split_k> [out.json]"""
import csv, glob, json, sys, collections

d, wgs = sys.argv[1], int(sys.argv[2])
M, N, K, split = [int(v) for v in sys.argv[3:7]]
rows = [r for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
sel = collections.defaultdict(list)
for r in rows:
    if "gemm_planes_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) == wgs:
        sel[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in sel.items()}
cd = lambda a, b: (a + b - 1) // b
# tile-K-steps of the group: dgrad [M x K_in] = dY[M x N_out] W (K loop over N_out), wgrad [N_out x K_in] (K loop over the M tokens)
steps_d = cd(M, 64) * cd(K, 64) * cd(N, 64)
steps_w = cd(N, 64) * cd(K, 64) * cd(M, 64)
n_mfma = (steps_d + steps_w) * 8 * 12
out = {"kernel": f"gemm_planes_kernel<3> x{wgs} (dgrad + wgrad of dY[{M}x{N}] against W[{N}x{K}], wgrad split-K {split})",
       "tile_k_steps": steps_d + steps_w, "mfma_instructions_analytic": n_mfma, "counters_avg_per_launch": avg}
print(json.dumps(out, indent=1))
if len(sys.argv) > 7:
    json.dump(out, open(sys.argv[7], "w"), indent=1)
