"""List, per kernel of a device-only assembly file (hipcc -S --cuda-device-only), every s_waitcnt with a vmcnt term that the
COMPILER inserted (i.e. not inside an inline-asm block), and instruction counts of interest.  Usage: isa_waits.py file.s [filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(\w+):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name or not name.startswith("_Z"):
        continue
    lines = body.split('\n')
    inasm = False
    cnt = {}
    waits = []
    for i, l in enumerate(lines):
        if 'ASMSTART' in l:
            inasm = True
        if 'ASMEND' in l:
            inasm = False
        t = l.strip()
        if t.startswith('s_waitcnt') and 'vmcnt' in t and not inasm:
            waits.append((i, t))
        for k in ('v_mfma', 'v_smfmac', 'buffer_load_dwordx4', 'ds_read_b128', 'ds_read_b64', 'ds_write_b64', 'ds_write_b128', 'global_store',
                  'global_load', 'buffer_load', 'buffer_store', 's_barrier', 'scratch_'):
            if t.startswith(k):
                cnt[k] = cnt.get(k, 0) + 1
    print(name[:90], "lines", len(lines))
    print("  ", cnt)
    for i, t in waits:
        print("   compiler wait @%d: %s" % (i, t))
