"""List the wait / barrier / branch skeleton of one kernel in a hipcc -S dump: python tools/isa_events.py FILE.s KERNEL [max]"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(_ZN\S*' + name + r'\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
body = m.group(2)
lines = body.split('\n')
print(name, len(lines), 'lines; vmcnt waits', len(re.findall(r's_waitcnt[^\n]*vmcnt', body)), 'lgkm waits', len(re.findall(r's_waitcnt[^\n]*lgkmcnt', body)),
      'barriers', body.count('s_barrier'), 'global_load', len(re.findall(r'global_load', body)), 'global_store', len(re.findall(r'global_store', body)),
      'ds_read', len(re.findall(r'ds_read', body)), 'ds_write', len(re.findall(r'ds_write', body)), 'scratch', body.count('scratch_'), 'mfma', body.count('v_mfma'))
n = 0
gl = 0
for i, l in enumerate(lines):
    l = l.strip()
    if 'global_load' in l: gl += 1
    if l.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_endpgm')) or l.startswith('.LBB'):
        print(f'{i}: {l.split(";")[0].strip()}   [global loads so far {gl}]')
        n += 1
        if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 200: break
