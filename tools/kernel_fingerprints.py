"""Static identity and resources of every kernel in the built library, from the library itself.

    python tools/kernel_fingerprints.py [libredmax_hip.so] [out.json]        (default out: redmax_amd/kernel_fingerprint.json)

llvm-objdump --offloading extracts the gfx950 code objects from the shared library, llvm-objdump -d disassembles them and
llvm-readelf --notes gives the kernel descriptors' metadata.  Per kernel symbol: instruction count, instruction-class histogram
(tools/isa_blocks.cls), a hash of the opcode sequence, VGPR / AGPR / SGPR counts, scratch (private segment) and static LDS bytes.
__graft_entry__.build() runs this after linking; bench.py compares the fingerprints of the kernels a workload launches with the
ones stored beside the roofline calibration (profiles/roofline_calibration.json): counter calibrations hold for the code they were
measured on and for nothing else.
"""
import collections
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LLVM = os.environ.get("RMX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
sys.path.insert(0, HERE)
from isa_blocks import cls  # noqa: E402


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool] + list(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return dict(zip(names, out))
        except (OSError, subprocess.CalledProcessError):
            continue
    return {n: n for n in names}


def code_objects(lib, workdir):
    dst = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], cwd=workdir, capture_output=True, check=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f and os.path.getsize(os.path.join(workdir, f)) > 0)


def disassemble(co):
    """{symbol: [instruction text, ...]} for every function of one code object"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if ins:
            cur.append(ins)
    return funcs


def metadata(co):
    """{symbol: {vgpr, agpr, sgpr, scratch_bytes, lds_bytes}} from the amdhsa.kernels notes"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out, cur = {}, None
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".private_segment_fixed_size": "scratch_bytes",
            ".group_segment_fixed_size": "lds_bytes", ".vgpr_spill_count": "vgpr_spills", ".sgpr_spill_count": "sgpr_spills"}
    block = {}
    for line in txt.split("\n"):
        s = line.strip()
        if s.startswith("- .agpr_count") or s.startswith("- .args"):
            if block.get("name"):
                out[block["name"]] = {k: v for k, v in block.items() if k != "name"}
            block = {}
            s = s[2:]
        m = re.match(r"^(\.\w+):\s+(.*)$", s)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == ".name":
            block["name"] = v
        elif k in keys:
            try:
                block[keys[k]] = int(v)
            except ValueError:
                pass
    if block.get("name"):
        out[block["name"]] = {k: v for k, v in block.items() if k != "name"}
    return out


def _one_code_object(co):
    part = {}
    meta = metadata(co)
    for sym, ins in disassemble(co).items():
        if sym not in meta:          # device functions that are not kernels (none today: everything is inlined)
            continue
        c = collections.Counter(cls(s) for s in ins)
        h = hashlib.sha256("\n".join(s.split()[0] for s in ins).encode()).hexdigest()[:16]
        part[sym] = dict({"instructions": len(ins), "classes": dict(sorted(c.items())), "opcode_sha16": h}, **meta[sym])
    return part


def fingerprints(lib):
    work = tempfile.mkdtemp(prefix="rmx_isa_")
    try:
        res = {}
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            for part in ex.map(_one_code_object, code_objects(lib, work)):
                res.update(part)
        names = demangle(list(res))
        for sym in res:
            res[sym]["name"] = names.get(sym, sym)
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "redmax_amd", "libredmax_hip.so")
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "redmax_amd", "kernel_fingerprint.json")
    fp = fingerprints(lib)
    json.dump(fp, open(out, "w"), indent=1, sort_keys=True)
    for sym in sorted(fp, key=lambda s: fp[s]["name"]):
        f = fp[sym]
        print("%-72s %6d instr  vgpr %3s agpr %3s scratch %5s B  lds %6s B  %s" % (f["name"][:72], f["instructions"], f.get("vgpr"), f.get("agpr"),
                                                                                   f.get("scratch_bytes"), f.get("lds_bytes"), f["opcode_sha16"]))


if __name__ == "__main__":
    main()
