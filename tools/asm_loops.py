"""Development: per-loop instruction histogram of one kernel in hipcc -S output.
usage: asm_loops.py file.s kernel_substring"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ': ' in l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
def klass(op):
    if op.startswith('v_') and not op.startswith('v_mov') : 
        if re.match(r'v_(rcp|rsq|log|exp|sqrt|sin|cos)', op): return 'trans'
        return 'valu'
    if op.startswith('v_mov'): return 'vmov'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return 'other'
loops = []
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch\w*\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            loops.append((labels[tgt], i, tgt))
def hist(a, b):
    h = collections.Counter()
    ops = collections.Counter()
    for l in body[a:b + 1]:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if not m or l.strip().startswith(('.', ';')): continue
        op = m.group(1)
        h[klass(op)] += 1
        ops[op] += 1
    return h, ops
print("whole:", dict(hist(0, len(body) - 1)[0]))
for a, b, t in sorted(loops, key=lambda x: x[0] - x[1])[:6]:
    h, ops = hist(a, b)
    print(f"loop {t} lines {a}-{b}: {dict(h)}")
    if '-v' in sys.argv: print("   ", ops.most_common(40))
