"""Development aid: libredmax_hip with the RMX_PART 4 object (the kernels around newton_pair, rmx_ct32.h) recompiled under extra flags,
as redmax_amd/variants/libredmax_hip_<name>.so; every other object comes from build/ (run __graft_entry__.build() first).
    python tools/pair_variants.py name1 "flags1" [name2 "flags2" ...]      e.g.  ph1 "-DRMX_TICK_PHASE=1"
tools/pair_bench.py times config 5 on the in-tree library and on every variant."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    args = sys.argv[1:]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", ge.CSRC,
            "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DRMX_NP=32", "-DRMX_PART=4"]
    vdir = os.path.join(ROOT, "build", "variants")
    odir = os.path.join(ROOT, "redmax_amd", "variants")
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    procs = []
    for name, flags in zip(args[0::2], args[1::2]):
        obj = os.path.join(vdir, "p4_%s.o" % name)
        fl = flags.split()
        if "--no-ilp" in fl:
            fl.remove("--no-ilp")
            b = [x for x in base if x not in ("-amdgpu-sched-strategy=max-ilp",)]
            b.remove("-mllvm")
        else:
            b = base
        procs.append((name, obj, subprocess.Popen([hipcc] + b + fl + ["-c", "-o", obj, ge.HIP_KERNEL_SRC])))
    others = sorted(os.path.join(ge.OBJ_DIR, f) for f in os.listdir(ge.OBJ_DIR) if f.endswith(".o") and f != "rmx_kernels_np32_p4.o")
    for name, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed for " + name)
        out = os.path.join(odir, "libredmax_hip_%s.so" % name)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, obj] + others)
        print(out)


main()
