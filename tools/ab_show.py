import json, sys
lib = None
for ln in open(sys.argv[1]):
    if ln.startswith('LIB'):
        lib = ln.split('/')[-1].strip()
    elif ln.startswith('{'):
        j = json.loads(ln)
        print(lib, j['n'], j['B'], '%.3e' % j['solves_per_s'], '%.2f' % j['us_per_step'])
