"""Per-basic-block instruction mix of one kernel in a hipcc -S listing.
usage: isa_blocks.py listing.s mangled_kernel_name [block_label_to_print ...]"""
import re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ':')
j = s.index('.Lfunc_end', i)
body = s[i:j]
blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', body)
tot = 0
for b in blocks:
    lines = b.split('\n')
    name = lines[0].split(':')[0] if lines[0].startswith('.LBB') else 'entry'
    ins = [l.strip().split()[0] for l in lines
           if l.startswith('\t') and not l.strip().startswith((';', '.'))]
    loop = 'Loop' in lines[0]
    cnt = lambda f: sum(1 for x in ins if f(x))
    print(f'{name:12s} n={len(ins):5d} valu={cnt(lambda x: x.startswith("v_")):5d} f64={cnt(lambda x: "f64" in x):4d} '
          f'cnd={cnt(lambda x: "cndmask" in x):3d} salu={cnt(lambda x: x.startswith("s_")):4d} '
          f'mem={cnt(lambda x: x.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_"))):3d} '
          f'lane={cnt(lambda x: "readlane" in x or "writelane" in x):3d} {"loop" if loop else ""}')
    tot += len(ins)
    if name in sys.argv[3:]:
        print(b)
print('total', tot)
