"""Per-slot instruction mix of a slot-scheduled kernel from hipcc's -S output (tools/asm_stats.sh):
    python tools/asm_slots.py gpurun_out/asm/pair_mlp_f16.s edge_transition_f16_kernelILb1E [mfmas_per_slot]
Prints the class counts between every n-th MFMA (a "slot") and the totals; v_accvgpr moves and s_nop are listed on their own."""
import collections
import re
import sys


def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith(('s_load', 's_buffer', 's_memtime')): return 'smem'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main():
    path, name = sys.argv[1], sys.argv[2]
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(name) + r'\w*:', l)][0]
    end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
    body = [l.strip() for l in lines[start:end] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    tot = collections.Counter(cls(l.split()[0]) for l in body)
    print('instructions', len(body), dict(tot))
    slots, cur, n = [], collections.Counter(), 0
    for l in body:
        k = cls(l.split()[0])
        cur[k] += 1
        if k == 'mfma':
            n += 1
            if n % per == 0:
                slots.append(cur)
                cur = collections.Counter()
    slots.append(cur)
    keys = ['valu', 'acc', 'lds', 'vmem', 'salu', 'smem', 'wait', 'nop', 'bar']
    print('slot ' + ' '.join(f'{k:>5}' for k in keys))
    for i, s in enumerate(slots):
        print(f'{i:4d} ' + ' '.join(f'{s.get(k, 0):5d}' for k in keys))


if __name__ == '__main__':
    main()
