"""Register / scratch / LDS use of the device kernels, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
    python tools/kernel_resources.py fastbox_amd/csrc/fb_fft_f32.hip [filter-substring ...] [-- extra hipcc flags]
"""
import re
import subprocess
import sys


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src, filters = args[0], args[1:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-c", src,
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:[^ ]+ )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void fb::", "")
        if filters and not all(f in name for f in filters):
            continue
        print("%-62s vgpr %4s agpr %3s sgpr %4s scratch %5s occ %s spill s/v %s/%s" % (
            name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"),
            r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))


if __name__ == "__main__":
    main()
