"""Build container: copy the summaries a GPU run of tools/collect_profiles.sh left under gpurun_out/prof_<tag>/ into
profiles/<tag>_* and record in profiles/current.json which files bench.py's `from_profiles` object refers to and the
commit the library was at when they were collected (the GPU box has no .git: stamp right after the run, before any
further commit).    python tools/stamp_profiles.py r02"""
import json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
for name in sorted(os.listdir(src)):
    if name.endswith((".csv", ".json", ".txt")):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "fastbox_amd", "bench.py"], capture_output=True, text=True).stdout.strip())
json.dump({"head": head + ("+uncommitted changes" if dirty else ""), "tag": tag,
           "pmc": "%s_pmc_fetch_write_summary.json" % tag,
           "kernel_stats": "%s_kernel_stats_streams1_pb64.csv" % tag,
           "kernel_stats_default_command": "%s_kernel_stats.csv" % tag,
           "kernel_stats_one_box": "%s_kernel_stats_streams1.csv" % tag,
           "kernel_stats_config3": "%s_kernel_stats_config3.csv" % tag,
           "note": "kernel_stats: rocprofv3 --kernel-trace --stats of `bench.py --streams 1 --plane-batch 64`, i.e. the launches of "
                   "the default command's roofline pass (one box at a time, the 64-plane batches two boxes per GPU use) -- the "
                   "average that bench.py's roofline.avg_launch_us has to agree with (HIP events add ~3 us of bracket per launch); "
                   "kernel_stats_default_command: the default command itself, whose average mixes that pass with launches that "
                   "shared the chip with the other box; kernel_stats_one_box: `--streams 1` with its own 128-plane batches; "
                   "pmc: the default command, dispatches serialised by the profiler; kernel_stats_config3: tools/config3_bench.py 512 (one box, "
                   "whole-box passes): k_rsd_turn and the other kernels of BASELINE configs[2]"},
          open(os.path.join(dst, "current.json"), "w"), indent=1)
print("stamped", head, "dirty" if dirty else "clean")
