"""Static check of the built gfx950 code object for the wait states the hardware does NOT interlock and the compiler cannot pad when
one side of the pair sits in inline asm (cdna_hip_programming.md 5.7 item 2):

  * a VALU write of a VGPR followed by a matrix instruction (v_mfma_*) reading it as SrcA / SrcB / SrcC: 2 wait states;
  * a VALU write of a VGPR followed by a DPP instruction reading it as its DPP source (src0): 2 wait states.

hipcc's hazard recognizer inserts these pads for instructions it emits itself; an `asm("v_fmac_f64_dpp ...")` is opaque to it, so
a matrix instruction taking an asm-written accumulator directly, or an asm DPP instruction whose source a compiler `v_mov` wrote the
instruction before, reads a stale value "now and then" (round 4 found one such case by accident).  This walks the disassembly of
EVERY kernel in remixt_amd/libremixt_hip.so -- compiler code and asm alike -- backwards from each consumer over all control-flow
predecessors (fall-through and every branch to a label) and reports any producer closer than the required states.  An `s_nop N`
counts N + 1 states, every other instruction 1.

    python tools/asm_hazards.py [library]        # prints violations, exit code 1 if any
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = 2

_REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs_of(operand):
    """Set of ('v'|'a', index) named by one operand string."""
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix='rmx_haz_')
    fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, lib])
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                           '--input=' + fat, '--output=' + co, '--unbundle'])
    text = subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', '--symbolize-operands', co]).decode()
    for f in (fat, co):
        os.remove(f)
    os.rmdir(tmp)
    return text


def split_operands(s):
    out, depth, cur = [], 0, ''
    for ch in s:
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur.strip()); cur = ''
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Ins(object):
    __slots__ = ('mn', 'ops', 'text', 'labels')

    def __init__(self, mn, ops, text):
        self.mn, self.ops, self.text, self.labels = mn, ops, text, []


def parse(text):
    """{kernel: [Ins]} with branch-target labels attached to the instruction they precede."""
    kernels, cur, pending = {}, None, []
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            name = m.group(1)
            if re.match(r'^L\d+$', name):
                pending.append(name)
            else:
                cur = kernels.setdefault(name, [])
                pending = []
            continue
        if cur is None or not line.startswith('\t'):
            continue
        body = line.split('//')[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        ins = Ins(parts[0], split_operands(parts[1]) if len(parts) > 1 else [], body)
        ins.labels, pending = pending, []
        cur.append(ins)
    return kernels


def valu_writes(ins):
    """VGPRs / AGPRs a vector-ALU instruction writes (empty for everything that is not a VALU write of a vector register)."""
    mn = ins.mn
    if not mn.startswith('v_') or mn.startswith('v_mfma') or mn.startswith('v_smfmac'):
        return set()
    if mn.startswith('v_cmp') or mn.startswith('v_readlane') or mn.startswith('v_readfirstlane') or mn.startswith('v_nop'):
        return set()
    w = regs_of(ins.ops[0]) if ins.ops else set()
    if mn.startswith('v_swap'):
        w |= regs_of(ins.ops[1])
    return w


def hazard_reads(ins):
    """Registers whose freshness the instruction depends on without a hardware interlock."""
    mn = ins.mn
    if mn.startswith('v_mfma') or mn.startswith('v_smfmac'):
        r = set()
        for op in ins.ops[1:4]:
            r |= regs_of(op)
        return r
    if '_dpp' in mn:
        # VOP1 / VOP2 DPP: src0 is the lane-permuted operand
        src0 = ins.ops[1].split()[0] if len(ins.ops) > 1 else ''
        return regs_of(src0)
    return set()


def states(ins):
    if ins.mn == 's_nop':
        return int(ins.ops[0], 0) + 1
    return 1


def check_kernel(name, code):
    label_at = {}
    for i, ins in enumerate(code):
        for l in ins.labels:
            label_at[l] = i
    branches_to = {}
    for i, ins in enumerate(code):
        if ins.mn.startswith('s_cbranch') or ins.mn == 's_branch':
            t = label_at.get(ins.ops[0]) if ins.ops else None
            if t is not None:
                branches_to.setdefault(t, []).append(i)
    found = []
    for i, ins in enumerate(code):
        need = hazard_reads(ins)
        if not need:
            continue
        # walk back over every predecessor path while fewer than REQUIRED states lie in between
        stack, seen = [(i, 0)], set()
        while stack:
            j, gap = stack.pop()
            preds = []
            if j > 0 and code[j - 1].mn not in ('s_branch', 's_endpgm', 's_setpc_b64'):
                preds.append(j - 1)
            preds += branches_to.get(j, [])
            for p in preds:
                if (p, gap) in seen:
                    continue
                seen.add((p, gap))
                hit = valu_writes(code[p]) & need
                if hit:
                    found.append((name, gap, code[p].text, ins.text))
                    continue
                g2 = gap + states(code[p])
                if g2 < REQUIRED:
                    stack.append((p, g2))
    return found


def check(lib=None):
    lib = lib or os.path.join(ROOT, 'remixt_amd', 'libremixt_hip.so')
    kernels = parse(disassemble(lib))
    found, counts = [], {'kernels': len(kernels), 'mfma': 0, 'dpp': 0}
    for name, code in kernels.items():
        counts['mfma'] += sum(1 for c in code if c.mn.startswith('v_mfma'))
        counts['dpp'] += sum(1 for c in code if '_dpp' in c.mn)
        found += check_kernel(name, code)
    return found, counts


if __name__ == '__main__':
    found, counts = check(sys.argv[1] if len(sys.argv) > 1 else None)
    print('%d kernels, %d matrix instructions, %d DPP instructions checked; %d violations' % (counts['kernels'], counts['mfma'], counts['dpp'], len(found)))
    for name, gap, prod, cons in found[:60]:
        print('  %s: %d state(s) between\n      %s\n      %s' % (name[:70], gap, prod, cons))
    sys.exit(1 if found else 0)
