"""Instruction mix of the MFMA-carrying basic blocks of a kernel, read from hipcc's
--save-temps assembly (maintainer aid for the kernel work; not part of the product).

usage: python tools/isa_loop_mix.py file.s <kernel-name-substring> [min_mfma]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op in ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_rsq_f32', 'v_sqrt_f32'):
        return 'valu_trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    text = open(path).read()
    funcs = re.split(r'\n(?=_Z[^\n]*:\s*;)', text)
    for f in funcs:
        name = f.split(':', 1)[0]
        if pat not in name:
            continue
        print('==', name)
        blocks, cur = [], ['entry', []]
        blocks.append(cur)
        for ln in f.split('\n'):
            m = re.match(r'^(\.LBB\d+_\d+):', ln)
            if m:
                cur = [m.group(1), []]
                blocks.append(cur)
            elif ln.startswith('\t') and not ln.startswith(('\t.', '\t;')):
                cur[1].append(ln.strip())
        for label, ins in blocks:
            n = sum(1 for i in ins if i.startswith('v_mfma'))
            if n < min_mfma:
                continue
            cls = collections.Counter(classify(i.split()[0]) for i in ins)
            print('  block', label, 'instrs', len(ins), dict(cls))
            ops = collections.Counter(i.split()[0] for i in ins)
            print('   ', ', '.join('{} {}'.format(k, v) for k, v in ops.most_common(40)))


if __name__ == '__main__':
    main()
