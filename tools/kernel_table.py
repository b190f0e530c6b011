"""Markdown table of the step / adjoint kernels' resources, read from redmax_amd/kernel_fingerprint.json (what __graft_entry__.build()
wrote for the library it linked): the one source of the register / scratch figures quoted in DESIGN.md.   python tools/kernel_table.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = (  # (name fragment, what it runs)
    ("k_step_bdf1_pair32", "configs[1] headline: full 32-link chain, BDF1, two points per front"),
    ("k_step_bdf1<32, false, false, true, 0>", "the one-point kernel of the same chain (RMX_PAIRC=0; bit-identity reference)"),
    ("k_step_bdf1<32, false, false, true, 16>", "the same chain, two wavefronts per rollout (RMX_PAIRC=0, 128..512 rollouts)"),
    ("k_step_bdf1<64, false, false, false, 18>", "configs[2]: full 64-node tree, two wavefronts per rollout (<= 512 rollouts), no energy record"),
    ("k_step_bdf1<64, false, false, false, 16>", "the same with the per-step energy record"),
    ("k_step_bdf1<64, false, false, false, 19>", "configs[2] at > 512 rollouts: constants in global memory, four wavefronts per CU"),
    ("k_ground32", "configs[4]: 32-link chain on frictional ground, BDF2, rollouts + cooperative groups in one launch"),
    ("k_step_pair<false>", "the same as separate launches: rollouts"),
    ("k_step_pair<true>", "... and cooperative groups"),
    ("k_adjoint_fwd<16, 1, true, true>", "configs[3]: adjoint BDF1 forward sweep, full 16-link chain, second wavefront for M, D (<= 512 rollouts)"),
    ("k_adjoint_fwd<16, 1, false, true>", "the same, one wavefront per rollout (larger batches)"),
    ("k_adjoint_bwd<16, 1, true>", "configs[3]: backward sweep"),
    ("k_adjoint_fwd<64, 1, false, false>", "adjoint forward, 33..64 nodes"),
    ("k_step_bdf1<32, true, false, false, 0>", "generic contact / Euler-chart kernel, <= 32 nodes, BDF1"),
    ("k_step_bdf2<32, true, false, false, 0>", "generic contact / Euler-chart kernel, <= 32 nodes, BDF2"),
    ("k_big_step", "trees of 65..256 nodes (one workgroup per rollout)"),
)


def main():
    fp = json.load(open(os.path.join(ROOT, "redmax_amd", "kernel_fingerprint.json")))
    print("| kernel | runs | VGPR+AGPR | scratch B/lane | SGPR spills | instructions |")
    print("|---|---|---|---|---|---|")
    for frag, what in ROWS:
        hit = [v for v in fp.values() if frag in v["name"]]
        for v in hit[:2 if "k_big_step" in frag else 1]:
            nm = v["name"].replace("void ", "").split("(")[0].replace("(anonymous namespace)::", "")
            print("| `%s` | %s | %d | %d | %d | %d |" % (nm, what, v["vgpr"], v["scratch_bytes"], v["sgpr_spills"], v["instructions"]))


main()
