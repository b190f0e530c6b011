"""Instruction mix per basic block of one kernel in a hipcc -S listing:  python tools/isa_blocks.py file.s mangled_prefix [minsize]"""
import collections
import re
import sys


def cls(s):
    op = s.split()[0]
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith(('v_fma_f64', 'v_fmac_f64')): return 'fma64_dpp' if 'row_newbcast' in s else 'fma64'
    if op.startswith(('v_mul_f64', 'v_add_f64')): return 'addmul64'
    if op.startswith(('v_min_f64', 'v_max_f64', 'v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64', 'v_ldexp_f64', 'v_fract_f64', 'v_trig', 'v_cmp_', 'v_cmpx')) and 'f64' in op: return 'other64'
    if 'readlane' in op or 'readfirstlane' in op: return 'readlane'
    if op.startswith('v_mov') and ('dpp' in op or 'row_' in s or 'quad_perm' in s): return 'movdpp'
    if op.startswith('v_permlane'): return 'permlane'
    if op.startswith('v_accvgpr'): return 'accvgpr'
    if op.startswith('v_cndmask'): return 'cndmask'
    if op.startswith('v_mov'): return 'mov'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('global', 'buffer', 'scratch', 'flat')): return 'vmem'
    return 'other'


def kernel_instructions(path, prefix):
    """the instruction lines (labels, comments and directives dropped) of the first function whose mangled name starts with prefix"""
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith(prefix) and (l.rstrip().endswith(':') or ': ;' in l)][0]
    end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end')][0]
    out = []
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((';', '.')) or re.match(r'^\.?LBB\d+_\d+:', s):
            continue
        out.append(s)
    return out


def fingerprint(path, prefix):
    """Static identity of a kernel's machine code: instruction counts per class and a hash of the opcode sequence.  bench.py compares
    the fingerprint of the library it runs (written by __graft_entry__.build()) with the one stored beside the roofline
    calibration: per-stage executed-instruction counts are only valid for the code they were measured on."""
    import hashlib
    ins = kernel_instructions(path, prefix)
    c = collections.Counter(cls(s) for s in ins)
    h = hashlib.sha256("\n".join(s.split()[0] for s in ins).encode()).hexdigest()[:16]
    return {"kernel": prefix, "instructions": len(ins), "classes": dict(sorted(c.items())), "opcode_sha16": h}


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    minsize = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(':') or (l.startswith(prefix) and ': ;' in l)][0]
    end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end')][0]
    blocks, cur = [], None
    for l in lines[start + 1:end]:
        s = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        if not s or s.startswith((';', '.')):
            continue
        if cur is None:
            cur = ['entry', []]
            blocks.append(cur)
        cur[1].append(s)
    tot = collections.Counter()
    for name, ins in blocks:
        c = collections.Counter(cls(s) for s in ins)
        tot.update(c)
        if len(ins) >= minsize:
            br = [s for s in ins if s.startswith(('s_cbranch', 's_branch'))]
            print("%-10s %5d  %s  -> %s" % (name, len(ins), ' '.join('%s=%d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])), ' | '.join(b.split()[-1] for b in br[-2:])))
    print("TOTAL %d  %s" % (sum(tot.values()), ' '.join('%s=%d' % kv for kv in sorted(tot.items(), key=lambda kv: -kv[1]))))


if __name__ == "__main__":
    main()
