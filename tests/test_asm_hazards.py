"""CPU: static audit of the built gfx950 code object for the two wait-state rules that are not interlocked by the hardware and that the
compiler cannot enforce across an inline-asm boundary (tools/asm_hazards.py): a vector-ALU write of a VGPR needs two
wait states before a matrix instruction reads it as SrcA / SrcB / SrcC, and before a DPP instruction reads it as its lane-permuted source.
The forward-backward kernels write accumulators with asm `v_fmac_f64_dpp` and feed matrix instructions from them; round 4 found a stale
operand of that class by accident.  The check walks the disassembly of every kernel of remixt_amd/libremixt_hip.so over all control-flow
predecessors -- the library that ships is the library that is checked."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

LIB = os.path.join(ROOT, 'remixt_amd', 'libremixt_hip.so')


@pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'), reason='no ROCm LLVM tools')
def test_no_unpadded_valu_write_before_a_matrix_or_dpp_read():
    import asm_hazards
    assert os.path.exists(LIB), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    found, counts = asm_hazards.check(LIB)
    # the kernels that carry the asm are in the library at all (an empty disassembly would pass vacuously)
    assert counts['kernels'] > 200 and counts['mfma'] > 400 and counts['dpp'] > 10000, counts
    assert not found, '\n'.join('%s: %d state(s) between [%s] and [%s]' % f for f in found[:20])


def test_the_checker_sees_a_planted_hazard():
    """The walker on a hand-written listing: a VALU write one state before a matrix read, before a DPP read, reached through a back edge,
    and the padded forms that are fine."""
    import asm_hazards
    bad = '''
0000000000000000 <k_bad>:
\tv_add_f64 v[4:5], v[0:1], v[2:3]            // 0
\ts_nop 0                                       // 4
\tv_mfma_f64_4x4x4_4b_f64 v[8:9], v[4:5], v[6:7], v[8:9]   // 8
\tv_mov_b32_e32 v10, v11                      // c
\tv_fmac_f64_dpp v[12:13], v[10:11], v[14:15] row_newbcast:1 row_mask:0xf bank_mask:0xf   // 10
\ts_endpgm
0000000000000100 <k_loop>:
0000000000000100 <L0>:
\tv_mov_b32_dpp v2, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
\ts_nop 1
\tv_mov_b32_e32 v3, v5
\ts_cbranch_scc1 L0
\ts_endpgm
0000000000000200 <k_good>:
\tv_add_f64 v[4:5], v[0:1], v[2:3]
\ts_nop 1
\tv_mfma_f64_4x4x4_4b_f64 v[8:9], v[4:5], v[6:7], v[8:9]
\tv_add_f64 v[20:21], v[0:1], v[2:3]
\tv_mfma_f64_4x4x4_4b_f64 v[8:9], v[30:31], v[6:7], v[8:9]
\tglobal_load_dwordx2 v[10:11], v[0:1], off
\tv_fmac_f64_dpp v[12:13], v[10:11], v[14:15] row_newbcast:1 row_mask:0xf bank_mask:0xf
\ts_endpgm
'''
    kernels = asm_hazards.parse(bad)
    assert sorted(kernels) == ['k_bad', 'k_good', 'k_loop']
    assert len(asm_hazards.check_kernel('k_bad', kernels['k_bad'])) == 2
    loop = asm_hazards.check_kernel('k_loop', kernels['k_loop'])
    assert len(loop) == 1 and 'v_mov_b32_e32 v3' in loop[0][2]          # around the back edge: v_mov, s_cbranch (1 state), DPP read
    assert asm_hazards.check_kernel('k_good', kernels['k_good']) == []
