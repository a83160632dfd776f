"""Register / scratch use of every kernel in ogmm_amd/csrc/build/*.o, read from the code objects' metadata notes (no GPU needed).
usage: python3 tools/kernel_resources.py [--spills]      (--spills: only kernels with spilled registers or scratch)
A run-time branch added to a hot kernel can push it over its register budget without any visible sign but these numbers (the fused Cout = 1 head
cost the default GEMM engine 87 spilled registers until it became its own instantiation)."""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
BUILD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ogmm_amd", "csrc", "build")


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.o")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], check=True, capture_output=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + dev],
                       check=True, capture_output=True)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", dev], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key in ("agpr_count", "args") and cur is not None and "name" in cur:          # a new kernel entry starts
            out.append(cur); cur = None
        if cur is None:
            cur = {}
        if key in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            cur[key] = val
    if cur and "name" in cur:
        out.append(cur)
    return out


def demangle(names):
    if not names:
        return names
    try:
        r = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
        return r.stdout.splitlines() if r.returncode == 0 else names
    except OSError:
        return names


def main():
    only_spills = "--spills" in sys.argv
    rows = []
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        try:
            ks = kernels_of(obj)
        except subprocess.CalledProcessError:
            continue          # a host-only object
        names = demangle([k["name"] for k in ks])
        for k, n in zip(ks, names):
            rows.append((os.path.basename(obj), n, int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0)), int(k.get("vgpr_spill_count", 0)),
                         int(k.get("private_segment_fixed_size", 0)), int(k.get("group_segment_fixed_size", 0))))
    print("%-24s %5s %5s %6s %8s %8s  kernel" % ("object", "vgpr", "agpr", "spills", "scratch", "lds"))
    for o, n, v, a, sp, sc, lds in rows:
        if only_spills and sp == 0 and sc == 0:
            continue
        print("%-24s %5d %5d %6d %8d %8d  %s" % (o, v, a, sp, sc, lds, n[:110]))
    return rows


if __name__ == "__main__":
    main()
