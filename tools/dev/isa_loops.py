"""Developer aid: per-loop instruction mix of one kernel in a hipcc -S listing (MFMAs, loads, LDS ops, scratch traffic, waits).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only x.hip -o x.s ;  python tools/dev/isa_loops.py x.s <kernel-substring>
"""
import re
import sys

text = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r'^_Z\S*:', l) and sub in l)
end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))   # (a kernel may hold several s_endpgm)
lines = text[start:end + 1]
print("kernel", lines[0][:100], "lines", len(lines))
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
cnt = lambda body, key: sum(key in x for x in body)
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        body = lines[a:i]
        print("loop %s [%d..%d]: mfma %d  global_load %d  ds_read %d  ds_write %d  scratch %d  accvgpr %d  valu(other) %d  salu %d  barrier %d" % (
            m.group(1), a, i, cnt(body, 'v_mfma'), cnt(body, 'global_load'), cnt(body, 'ds_read'), cnt(body, 'ds_write'),
            cnt(body, 'scratch_'), cnt(body, 'v_accvgpr'),
            sum(1 for x in body if re.match(r'\s+v_', x) and 'v_mfma' not in x and 'v_accvgpr' not in x),
            sum(1 for x in body if re.match(r'\s+s_', x) and 's_waitcnt' not in x and 's_nop' not in x), cnt(body, 's_barrier')))
        print("    waits:", [x.strip() for x in body if 's_waitcnt' in x])
