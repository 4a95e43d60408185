"""Instruction mix of the walk kernels' 8-entry basic block (development aid): python tools/hot_loop.py /tmp/pileup.s [print]"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
for fn in ("_ZN3lsg12k_walk_blockENS_9CountArgsE", "_ZN3lsg13k_pileup_waveENS_9CountArgsE"):
    st = [i for i, l in enumerate(L) if l.startswith(fn + ':')][0]
    en = [i for i in range(st, len(L)) if 's_endpgm' in L[i]][0]
    blocks = []; cur = []
    for l in L[st:en]:
        if re.match(r'^\.LBB', l): blocks.append(cur); cur = []
        cur.append(l)
    blocks.append(cur)
    for b in blocks:
        nl = sum('buffer_load_ushort' in x for x in b); nd = sum('ds_add_u32' in x for x in b)
        if nl >= 8 and nd >= 8:
            ins = [re.sub(r'\s*;.*', '', x).strip() for x in b if x.strip() and not x.strip().startswith(('.', ';'))]
            ops = [i.split()[0] for i in ins]
            v = sum(i.startswith('v_') for i in ops); sc = sum(i.startswith('s_') for i in ops)
            print(fn[8:22], "block", b[0][:10], "len", len(ops), "VALU", v, "SALU", sc, "loads", nl, "ds_add", nd)
            print("  ", collections.Counter(i for i in ops if i.startswith('v_')).most_common())
            print("  ", collections.Counter(i for i in ops if i.startswith('s_')).most_common())
            if len(sys.argv) > 2: print("\n".join(ins))
    for l in L[en:en + 80]:
        if re.search(r'NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize', l): print("  ", l.strip())
