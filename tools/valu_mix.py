"""The instruction mix of the Smith-Waterman passes' sweep loops, from the compiler's assembly: python tools/valu_mix.py [out.json]
For sw_score_kernel<false> / sw_trace_kernel<false> (the launches of the pairs that fit the staging area: the packed 16-bit sweeps) the loop that holds the
packed arithmetic is located (the smallest loop with more than half of the kernel's v_pk_* instructions) and its vector instructions counted by opcode.
bench.py weights the VALU issue ceiling with it: tools/micro/valu_rate measures 4.16 cycles per wave64 instruction for the packed-16 / DPP / three-operand
class and 2.27 for plain 32-bit VOP1 / VOP2 (v_mov_b32, v_add_u32, v_and_b32 ...) - a ceiling that prices every instruction at 4.16 puts a kernel with
a tenth of two-cycle instructions above 1.0 (round 5's line: 1.0026)."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# opcodes tools/micro/valu_rate times at ~2.27 cycles per wave64 instruction and SIMD (profiles/r0N_valu_rate.txt); everything else counts 4.16
TWO_CYCLE = ('v_mov_b32_e32', 'v_add_u32_e32', 'v_sub_u32_e32', 'v_subrev_u32_e32', 'v_and_b32_e32', 'v_or_b32_e32', 'v_xor_b32_e32', 'v_lshlrev_b32_e32', 'v_lshrrev_b32_e32',
             'v_ashrrev_i32_e32', 'v_not_b32_e32', 'v_fma_f32', 'v_add_f32_e32', 'v_mul_f32_e32')


def loops_of(body):
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    out = []
    for k, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            out.append((labels[m.group(1)], k))
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'sw.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-fast-math', '-S', '--cuda-device-only', '-o', asm,
                               os.path.join(ROOT, 'peppan_amd', 'csrc', 'sw.hip')], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split('\n')
    res = {}
    for i0, l in enumerate(lines):
        m = re.match(r'^_ZN\S*?(sw_score_kernel|sw_trace_kernel)ILb0E\S*:', l)
        if not m:
            continue
        i1 = next(j for j in range(i0, len(lines)) if 's_endpgm' in lines[j])
        body = lines[i0:i1]
        is_v = lambda s: re.match(r'\tv_', s) is not None
        pk_all = sum(1 for s in body if re.match(r'\tv_pk_', s))
        best = None
        for a, b in loops_of(body):
            pk = sum(1 for s in body[a:b + 1] if re.match(r'\tv_pk_', s))
            if pk * 2 > pk_all and (best is None or b - a < best[1] - best[0]):
                best = (a, b)
        a, b = best
        ops = collections.Counter(s.split()[0] + ('_dpp' if ('row_' in s or 'wave_' in s or 'quad_perm' in s) and not s.split()[0].endswith('_dpp') else '') for s in body[a:b + 1] if is_v(s))
        n = sum(ops.values())
        two = sum(c for o, c in ops.items() if o in TWO_CYCLE)
        res[m.group(1)] = {'loop_valu_instructions': n, 'two_cycle_class': two, 'two_cycle_frac': two / float(n), 'lds_instructions': sum(1 for s in body[a:b + 1] if re.match(r'\tds_', s)),
                           'salu_instructions': sum(1 for s in body[a:b + 1] if re.match(r'\ts_', s)), 'opcodes': dict(ops.most_common())}
    text = json.dumps({'source': 'tools/valu_mix.py: hipcc -S of peppan_amd/csrc/sw.hip (gfx950, -O3), the sweep loop of the <false> instances', 'two_cycle_opcodes': list(TWO_CYCLE), 'kernels': res}, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main()
