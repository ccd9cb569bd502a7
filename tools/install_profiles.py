#!/usr/bin/env python3
"""Copy the summaries tools/refresh_profiles.sh left under gpurun_out/<tag> into profiles/ with the round's prefix.

  python tools/install_profiles.py gpurun_out/<tag> r02
"""
import glob, os, shutil, sys
src, pre = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(os.path.join(dst, f"{pre}_bench"), exist_ok=True)
copied = []
def cp(a, b):
    if os.path.exists(a):
        shutil.copy(a, os.path.join(dst, b)); copied.append(b)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    cp(f, os.path.join(f"{pre}_bench", os.path.basename(f)))
cp(os.path.join(src, "bench_c3.json"), f"{pre}_bench_default.json")
cp(os.path.join(src, "eval", "kernel_stats.csv"), f"{pre}_kernel_stats_bf16x3.csv")
cp(os.path.join(src, "eval", "kernel_table.txt"), f"{pre}_kernel_table.txt")
for k in ("conv_tangent_bf16x3", "conv_tangent_bf16x3_live", "gram_cholesky", "acl_tangent", "conv_tangent_f16x3", "conv_tangent_thin"):
    cp(os.path.join(src, "eval", f"pmc_{k}.json"), f"{pre}_pmc_{k}.json")
cp(os.path.join(src, "eval", "pmc_conv_tangent_bf16x3.json"), "pmc_conv_tangent_bf16x3.json")   # bench.py's roofline.traffic reads this one
cp(os.path.join(src, "train", "kernel_stats.csv"), f"{pre}_train_kernel_stats.csv")
cp(os.path.join(src, "wgrad", "pmc_conv_wgrad.json"), f"{pre}_pmc_conv_wgrad.json")
cp(os.path.join(src, "train_step_b64.txt"), f"{pre}_train_step_b64.txt")
cp(os.path.join(src, "c5train", "kernel_stats.csv"), f"{pre}_c5_train_kernel_stats.csv")
if os.path.exists(os.path.join(src, "c5_train_variants.txt")):
    lines = [l for l in open(os.path.join(src, "c5_train_variants.txt")) if l.startswith("C5 train")]
    open(os.path.join(dst, f"{pre}_c5_train_variants.txt"), "w").write(
        "# python tools/exp_c5_train.py --modes 16,32,full ; --modes 32 --B 256   (one MI355X: CIFAR d=128, train-mode Hutchinson S=4 + CG, fwd + loss.backward() + Adam;\n"
        "# mode = column slots of the low-rank sweep (HUTCH_LOWRANK_NC) or the d-column backward of rounds 1-2)\n" + "".join(lines))
    copied.append(f"{pre}_c5_train_variants.txt")
cp(os.path.join(src, "mfma_sustained.txt"), f"{pre}_mfma_sustained.txt")
cp(os.path.join(src, "stage_table.txt"), f"{pre}_stage_table.txt")
cp(os.path.join(src, "c5_train_step.txt"), f"{pre}_c5_train_step.txt")
cp(os.path.join(src, "f16x3_item_sizes.txt"), f"{pre}_f16x3_item_sizes.txt")
if os.path.exists(os.path.join(src, "gpu_tests.log")):
    open(os.path.join(dst, f"{pre}_gpu_tests_summary.txt"), "w").write("".join(open(os.path.join(src, "gpu_tests.log")).readlines()[-3:]))
    copied.append(f"{pre}_gpu_tests_summary.txt")
print("\n".join(copied))
