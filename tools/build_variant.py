"""Development aid: build libredmax_hip with ONE kernel translation unit recompiled under extra flags, as
redmax_amd/variants/libredmax_hip_<name>.so (the other objects come from build/, i.e. run __graft_entry__.build() first).

    python tools/build_variant.py <name> [--np 32] [--part 0] [--no-ilp] -- <extra hipcc flags, e.g. -DRMX_VAR_X=1 -mllvm -foo>

tools/variant_bench.py times every variant in that directory on the GPU box (the .so files travel with the snapshot)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--np", type=int, default=32)
    ap.add_argument("--part", type=int, default=0)
    ap.add_argument("--no-vform", action="store_true")
    ap.add_argument("--no-ilp", action="store_true", help="drop -amdgpu-sched-strategy=max-ilp where the in-tree build uses it (part 1, 64-lane objects)")
    ap.add_argument("--asm", action="store_true", help="also write the device assembly to build/isa/var_<name>.s")
    a = ap.parse_args(argv)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", ge.CSRC]
    ilp = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if ((a.part == 1 or a.np == 64 or (a.np == 16 and a.part in (0, 8))) and not a.no_ilp) else []
    if a.np >= 32 and not a.no_vform:
        ilp += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    vdir = os.path.join(ROOT, "build", "variants")
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, "%s.o" % a.name)
    part_flags = {3: ["-DRMX_GLOBAL_CONSTS"], 6: ["-DRMX_W2=1", "-DRMX_SYNC()=rmx_wave_sync()", "-DRMX_CONSTS(sAcc,n,NP)=(rmx_smem_base()+acc_doubles((n),(NP)))"], 5: ["-DRMX_W2=1", "-DRMX_SYNC()=rmx_wave_sync()", "-DRMX_CONSTS(sAcc,n,NP)=(rmx_smem_base()+acc_doubles((n),(NP)))"]}.get(a.part, [])
    if a.part in (4, 7, 8) and not any(x.startswith("-DRMX_SYNC") for x in extra):
        part_flags = ["-DRMX_SYNC()=rmx_lane_sync()"]      # (the in-tree default of that part, __graft_entry__._build_hip)
    if any(x.startswith("-DRMX_SYNC") for x in extra):      # (a variant's own synchronisation macro replaces the part's)
        part_flags = [x for x in part_flags if not x.startswith("-DRMX_SYNC")]
    tu = ["-DRMX_NP=%d" % a.np, "-DRMX_PART=%d" % a.part] + part_flags + [ge.HIP_KERNEL_SRC]
    procs = [subprocess.Popen([hipcc] + flags + ilp + extra + ["-c", "-o", obj] + tu)]
    if a.asm:
        os.makedirs(os.path.join(ROOT, "build", "isa"), exist_ok=True)
        procs.append(subprocess.Popen([hipcc] + flags + ilp + extra + ["-S", "--cuda-device-only", "-o", os.path.join(ROOT, "build", "isa", "var_%s.s" % a.name)] + tu,
                                      stderr=subprocess.DEVNULL))
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("hipcc failed")
    objs = [os.path.join(ge.OBJ_DIR, "redmax_hip.o"), os.path.join(ge.OBJ_DIR, "rmx_big.o")]
    for n in ge.HIP_NPS:
        for part in (0, 1, 2):
            if part == 2 and n < 16:
                continue
            objs.append(obj if (n == a.np and part == a.part) else os.path.join(ge.OBJ_DIR, "rmx_kernels_np%d_p%d.o" % (n, part)))
    for n, part in ((64, 3), (32, 4), (64, 5), (32, 6), (32, 7), (16, 8)):      # the one-size parts
        objs.append(obj if (n == a.np and part == a.part) else os.path.join(ge.OBJ_DIR, "rmx_kernels_np%d_p%d.o" % (n, part)))
    out = os.path.join(ROOT, "redmax_amd", "variants", "libredmax_hip_%s.so" % a.name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
    print(out)


main()
