#!/usr/bin/env python3
"""developer tool: per-basic-block issue-cost estimate of a gfx950 kernel from hipcc's -S output, with the per-instruction
costs measured by tools/ubench/valu_rate.hip and lds_rate.hip (cycles of the SIMD / of the CU's LDS pipe per wave64
instruction, at >= 2 waves per SIMD):  fma/mul/add/sub f32, add/sub/and/or/xor/mov/ashr b32, cndmask: 2.3;  the same with
an SGPR source, min/max/med3, shifts, mul24/mad24, bfe, cvt, cmp, DPP, readlane, packed f32: 4.1;  rcp/sqrt/exp: 8.2.
usage: asm_cost.py file.s kernel_substring [min_cost]"""
import re, sys
CHEAP = ('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32',
         'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_mov_b32', 'v_ashrrev_i32', 'v_cndmask_b32', 'v_not_b32')
TRANS = ('v_rcp_', 'v_sqrt_', 'v_exp_', 'v_log_', 'v_rsq_', 'v_sin_', 'v_cos_')
LDS = {'ds_read_b32': 2.5, 'ds_read_u16': 2.5, 'ds_read_u8': 2.5, 'ds_read_b64': 2.5, 'ds_read2_b32': 5, 'ds_read_b128': 4.7, 'ds_read2_b64': 8,
       'ds_read_b96': 8, 'ds_write_b32': 4.4, 'ds_write_b16': 4.4, 'ds_write_b8': 4.4, 'ds_write_b64': 6, 'ds_write2_b32': 6, 'ds_write_b128': 13,
       'ds_write2_b64': 13, 'ds_write_b96': 10, 'ds_min_u64': 6.6, 'ds_min_rtn_u64': 6.6, 'ds_max_rtn_i32': 4.4, 'ds_max_i32': 4.4,
       'ds_add_f32': 100, 'ds_add_u32': 4.4, 'ds_add_rtn_u32': 4.4, 'ds_bpermute_b32': 4.4, 'ds_swizzle_b32': 4.4}
def cost(ins):
    op = ins.split()[0]
    base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if op.startswith('ds_'):
        return ('lds', LDS.get(base, 4.4))
    if not op.startswith('v_'):
        return ('other', 0.0)
    if any(base.startswith(t) for t in TRANS):
        return ('valu', 8.2)
    if base in CHEAP and 'dpp' not in op and ' row_' not in ins:
        # an SGPR source makes it full cost (vcc as the cndmask selector does not)
        ops = ins.split(None, 1)[1] if ' ' in ins else ''
        srcs = ops.split(',')[1:]
        if base == 'v_cndmask_b32':
            srcs = srcs[:2]
        if any(re.match(r'\s*-?\|?s\d+|\s*-?\|?s\[', s_) for s_ in srcs):
            return ('valu', 4.1)
        return ('valu', 2.3)
    return ('valu', 4.1)
def main():
    s = open(sys.argv[1]).read()
    kn = sys.argv[2]
    minc = float(sys.argv[3]) if len(sys.argv) > 3 else 60
    m = re.search(r'\n(_Z\w*' + re.escape(kn) + r'\w*):', s)
    i = m.start(1)
    f = s[i:s.index('s_endpgm', i) + 10]
    blocks = []; cur = [m.group(1), []]; blocks.append(cur)
    for l in f.splitlines():
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            cur = [mm.group(1), []]; blocks.append(cur)
        elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
            cur[1].append(l.split(';')[0].strip())
    tv = tl = 0
    for name, ins in blocks:
        v = sum(c for k, c in map(cost, ins) if k == 'valu'); l_ = sum(c for k, c in map(cost, ins) if k == 'lds')
        nv = sum(1 for x in ins if x.startswith('v_')); nl = sum(1 for x in ins if x.startswith('ds_'))
        tv += v; tl += l_
        if v + l_ >= minc:
            full = [x.split()[0] for x in ins if x.startswith('v_') and cost(x)[1] >= 4]
            from collections import Counter
            print('%-12s n=%3d valu %3d (%5.0f cyc) lds %2d (%4.0f cyc)  full-cost: %s' % (name, len(ins), nv, v, nl, l_, dict(Counter(full).most_common(8))))
    print('static total: valu %.0f lds %.0f' % (tv, tl))
main()
