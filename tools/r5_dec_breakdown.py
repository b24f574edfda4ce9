"""Per-tile instruction breakdown of the fused decoder backward kernel (dec_bwd_kernel<UT=4, FS, GIN=false, BT=2>: U = 50, rows >= 2^19 -
the T3 step at C3): every instruction of the 16-row tile loop attributed to the SOURCE LINE it was generated from (the translation unit is
compiled once more with -gline-tables-only; llvm-objdump -l) and the lines grouped by what they do.  Run here (no GPU needed):
    python tools/r5_dec_breakdown.py > profiles/r05_decoder_tile_breakdown.txt
"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import erratum_scan as E
SRC = os.path.join(ROOT, 'vmp-for-svae_amd', 'csrc', 'vmp_decoder.hip')
KERNEL = sys.argv[1] if len(sys.argv) > 1 else 'dec_bwd_kernelILi4ELb1ELb0ELi2E'
tmp = tempfile.mkdtemp()
obj = os.path.join(tmp, 'dec.o')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-gline-tables-only', '-Wno-unused-variable',
                       '-Wno-unused-but-set-variable', '-c', SRC, '-o', obj], stderr=subprocess.DEVNULL)
img = E.code_objects(open(obj, 'rb').read())[0]
co = os.path.join(tmp, 'dec.co')
open(co, 'wb').write(img)
txt = subprocess.run([E.OBJDUMP, '-d', '-l', '--no-show-raw-insn', co], check=True, capture_output=True, text=True).stdout.splitlines()
# ---- the kernel's instructions with (file, line)
ins, cur, on = [], ('?', 0), False
for ln in txt:
    m = re.match(r'^[0-9a-f]+ <(.+)>:$', ln)
    if m:
        on = KERNEL in m.group(1)
        continue
    if not on:
        continue
    m = re.match(r'^; (.+):(\d+)$', ln)
    if m:
        cur = (os.path.basename(m.group(1)), int(m.group(2)))
        continue
    if ln.startswith('\t') or ln.startswith(' '):
        body = ln.split('//')
        op = body[0].split()
        if not op:
            continue
        addr = int(body[1].split(':')[0].strip(), 16) if len(body) > 1 else None
        ins.append((addr, op[0], body[0].strip(), cur))
# ---- the tile loop: the LAST backward branch whose span contains the most MFMAs
addr_ix = {a: i for i, (a, _, _, _) in enumerate(ins) if a is not None}
best = None
for i, (a, op, text, _) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        m = re.search(r'<.*\+0x([0-9a-f]+)>', txt_line := text) or None
        # objdump prints the target as a comment we stripped; recompute from the simm16 operand
        off = int(text.split()[1])
        if off >= 32768:
            tgt = a + 4 + (off - 65536) * 4
            if tgt in addr_ix:
                j = addr_ix[tgt]
                nm = sum(1 for k in range(j, i) if ins[k][1].startswith('v_mfma'))
                if best is None or nm > best[0]:
                    best = (nm, j, i)
nm, j0, j1 = best
loop = ins[j0:j1 + 1]
src = open(SRC).read().splitlines()
def fn_of(file, line):
    """category of a source line"""
    if file == 'vmp_common.h':
        return 'bf16 splitting (v = h + m + l)' if 30 <= line <= 46 else 'common helpers (row sums ...)'
    if file != 'vmp_decoder.hip':
        return 'other file'
    L = line
    def within(name_start, name_end):
        return name_start <= L <= name_end
    if 137 <= L <= 186: return 'tanh / softplus / sigmoid evaluation'
    if 364 <= L <= 381: return 'tile inputs: row -> (cell, n) map and loads'
    if 382 <= L <= 397: return 'bf16 splitting (v = h + m + l)'
    if 354 <= L <= 363: return 'MFMA + operand fetch from the weight images (ds_read_b128)'
    if 399 <= L <= 479: return 'MFMA + operand fetch from the weight images (ds_read_b128)'
    if 480 <= L <= 516: return 'forward recompute glue (bias loads, ones unit)'
    if 575 <= L <= 627: return '(unit,row) <-> (row,unit) transposes through the LDS scratch'
    if 687 <= L <= 722: return 'tile inputs: row -> (cell, n) map and loads'
    if 723 <= L <= 736: return '(unit,row) <-> (row,unit) transposes through the LDS scratch'
    if 746 <= L <= 777: return 'reconstruction term: log-likelihood value and output gradients'
    if 778 <= L <= 812: return 'MFMA + operand fetch from the weight images (ds_read_b128)'
    if 813 <= L <= 821 or 852 <= L <= 859: return 'tanh derivative (1 - h^2) and bias-gradient sums'
    if 822 <= L <= 851 or 860 <= L <= 888: return 'MFMA + operand fetch from the weight images (ds_read_b128)'
    if 713 <= L <= 722: return 'tile inputs: row -> (cell, n) map and loads'
    return 'loop control / other'
def kind(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): return 'VMEM'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'): return 'wait/nop'
    return 'SALU'
tab = collections.defaultdict(collections.Counter)
ops = collections.defaultdict(collections.Counter)
for a, op, text, (f, l) in loop:
    c = fn_of(f, l)
    tab[c][kind(op)] += 1
    if kind(op) == 'VALU':
        ops[c][op.replace('_e32', '').replace('_e64', '')] += 1
tot = collections.Counter()
for c in tab:
    tot.update(tab[c])
print('Fused decoder backward, tile loop of %s (one 16-row tile per iteration and wave; U = 50 -> UT = 4 unit tiles, 2-term operands on the' % KERNEL)
print('backward data path).  Instructions of ONE iteration, attributed to the source lines they were generated from (llvm-objdump -l on a')
print('-gline-tables-only build of csrc/vmp_decoder.hip; tools/r5_dec_breakdown.py).  %d instructions: %s' % (len(loop), dict(tot)))
print()
print('%-78s %6s %6s %6s %6s %6s %8s' % ('category', 'VALU', 'MFMA', 'LDS', 'VMEM', 'SALU', 'wait/nop'))
for c, cnt in sorted(tab.items(), key=lambda kv: -kv[1]['VALU']):
    print('%-78s %6d %6d %6d %6d %6d %8d   (%4.1f %% of the VALU instructions)' % (c, cnt['VALU'], cnt['MFMA'], cnt['LDS'], cnt['VMEM'], cnt['SALU'], cnt['wait/nop'],
                                                                                   100.0 * cnt['VALU'] / max(1, tot['VALU'])))
print()
print('VALU opcodes per category (top 8):')
for c, cnt in sorted(ops.items(), key=lambda kv: -sum(kv[1].values())):
    print('  %-76s %s' % (c, ', '.join('%s %d' % kv for kv in cnt.most_common(8))))
