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
                       '-Wno-unused-but-set-variable'] + os.environ.get('EXTRA', '').split() + ['-c', SRC, '-o', obj], stderr=subprocess.DEVNULL)
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
# categories by ANCHOR lines found in the source text (no hard-coded line numbers): a line belongs to the last anchor at or before it
ANCHORS = [
    ('__device__ __forceinline__ float rcp_f', 'tanh / softplus / sigmoid evaluation'),
    ('__device__ __forceinline__ int kslot_unit', 'loop control / other'),
    ('__device__ __forceinline__ f32x4 lds4', 'MFMA + operand fetch from the weight images (ds_read_b128)'),
    ('struct RowMap {', 'tile inputs: row -> (cell, n) map and loads'),
    ('__device__ __forceinline__ void split_tiles', 'bf16 splitting (v = h + m + l)'),
    ('__device__ __forceinline__ void gemm_units', 'MFMA + operand fetch from the weight images (ds_read_b128)'),
    ('__device__ __forceinline__ void dec_forward_tile', 'forward recompute glue (bias loads, ones unit)'),
    ('__device__ __forceinline__ void wave_lds_order', '(unit,row) <-> (row,unit) transposes through the LDS scratch'),
    ('void dec_bwd_kernel(DecArgs a) {', 'loop control / other'),
    ('    struct TileIn {', 'tile inputs: row -> (cell, n) map and loads'),
    ('        unsigned xs[3];', '(unit,row) <-> (row,unit) transposes through the LDS scratch'),
    ('        f32x4 h0[UT], h1[UT], O;', 'forward recompute glue (bias loads, ones unit)'),
    ('        // ---- reconstruction term: gradients w.r.t. the output slots', 'reconstruction term: log-likelihood value and output gradients'),
    ('        // ---- dh1 = W2 . dO', 'MFMA + operand fetch from the weight images (ds_read_b128)'),
    ('        // ---- through tanh of layer 1', 'tanh derivative (1 - h^2) and bias-gradient sums'),
    ('        // ---- dh0 = W1 . dh1pre', 'MFMA + operand fetch from the weight images (ds_read_b128)'),
    ('        // ---- through tanh of layer 0', 'tanh derivative (1 - h^2) and bias-gradient sums'),
    ('        // ---- dx = W0 . dh0pre', 'MFMA + operand fetch from the weight images (ds_read_b128)'),
    ('    // ---- reduce the per-wave accumulators through LDS', 'loop control / other'),
]
_anch = []
for needle, cat in ANCHORS:
    hits = [i + 1 for i, l in enumerate(src) if l.startswith(needle) or (needle.startswith(' ') and l.rstrip() == needle.rstrip()) or needle in l and needle.startswith('void ')]
    if not hits:
        raise SystemExit('anchor not found in the source: %r' % needle)
    _anch.append((hits[0], cat))
_anch.sort()
common = open(os.path.join(os.path.dirname(SRC), 'vmp_common.h')).read().splitlines()
_split0 = next(i + 1 for i, l in enumerate(common) if 'void split_bf16(' in l)
_cvt0 = next((i + 1 for i, l in enumerate(common) if 'cvt_pk_bf16(' in l and '__device__' in l), None)
def fn_of(file, line):
    """category of a source line"""
    if file == 'vmp_common.h':
        if _split0 - 2 <= line <= _split0 + 12 or (_cvt0 is not None and _cvt0 - 1 <= line <= _cvt0 + 8):
            return 'bf16 splitting (v = h + m + l)'
        return 'common helpers (row sums ...)'
    if file != 'vmp_decoder.hip':
        return 'library math inlined from other headers (expf, logf ...)'
    cat = 'loop control / other'
    for l0, c in _anch:
        if l0 <= line:
            cat = c
    return cat
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
print(os.environ.get('TITLE', ''))
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
