"""Registers and scratch of every kernel in the built objects (celerite2_amd/build/*.o): lists the kernels that use more than 256
registers (arch + accumulation: ONE wavefront per SIMD) or any scratch -- intended for the fused log-likelihood pairs
(__launch_bounds__(64, 1)), an accident anywhere else (round 6: k_cols_walk<.., 0> at 442 registers, profiles/r06_large_nrhs.md).
Usage: python tools/kernel_regs.py [min registers, default 257]"""
import glob, os, re, shutil, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
llvm = "/opt/rocm/lib/llvm/bin"
lim = int(sys.argv[1]) if len(sys.argv) > 1 else 257
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for o in sorted(glob.glob(os.path.join(root, "celerite2_amd", "build", "c2_*.o"))):
        if re.search(r"_[a-z0-9]+\.o$", os.path.basename(o)) and not os.path.exists(os.path.join(root, "celerite2_amd", "csrc", os.path.basename(o)[:-2] + ".hip")):
            continue   # objects of A/B builds
        # (llvm-objdump --offloading writes the bundle's members next to its input: work on a copy)
        cp = os.path.join(tmp, os.path.basename(o))
        shutil.copy(o, cp)
        subprocess.run([llvm + "/llvm-objdump", "--offloading", cp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = glob.glob(cp + ".*hipv4-amdgcn*")
        if not cos: continue   # (host code only)
        co = cos[0]
        out = subprocess.run([llvm + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for b in re.split(r"\n\s*- \.agpr_count:", out)[1:]:
            ag = int(b.split("\n")[0].strip())
            name = re.search(r"\.name:\s*(\S+)", b).group(1)
            vg = int(re.search(r"\.vgpr_count:\s*(\d+)", b).group(1))
            sc = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", b).group(1))
            rows.append((os.path.basename(o)[:-2], name, vg, ag, sc))
names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(len(rows), "kernels; those with >= %d registers or scratch:" % lim)
for r, n in sorted(zip(rows, names), key=lambda x: (x[0][0], -x[0][2])):
    if r[2] >= lim or r[4] > 0:
        print("%-20s %-110s registers %4d (accumulation %3d) scratch %d B" % (r[0], n.split("(")[0].replace("void ", "")[:110], r[2], r[3], r[4]))
