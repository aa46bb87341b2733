import re,collections,sys
lines=open(sys.argv[1]).read().split('\n')
cnt=collections.defaultdict(collections.Counter)
cur=0
for ln in lines:
    if ln.startswith('.LBB') or ln.startswith('; %bb.'):
        m=re.search(r'Depth=(\d+)', ln)
        cur=int(m.group(1)) if m else 0
        continue
    m=re.search(r'^\s+;.*Depth=(\d+)', ln)
    if m and ('Loop Header' in ln or 'Inner Loop' in ln):
        cur=max(cur,int(m.group(1))); continue
    t=ln.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cnt[cur][t.split()[0]]+=1
for dpt in sorted(cnt):
    cats=collections.Counter()
    for op,c in cnt[dpt].items():
        if op.startswith('v_readlane') or op.startswith('v_writelane'): cats['lane-spill']+=c
        elif '_f64' in op: cats['f64']+=c
        elif op.startswith('v_cndmask'): cats['cndmask']+=c
        elif op.startswith('v_cmp'): cats['v_cmp']+=c
        elif op.startswith('v_mov'): cats['v_mov']+=c
        elif op.startswith('v_'): cats['v_other']+=c
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): cats['branch']+=c
        elif op.startswith('s_load'): cats['s_load']+=c
        elif op.startswith('s_'): cats['s_other']+=c
        elif op.startswith('ds_'): cats['ds']+=c
        elif op.startswith('global') or op.startswith('buffer') or op.startswith('flat'): cats['vmem']+=c
        else: cats[op]+=c
    print("depth",dpt,"total",sum(cnt[dpt].values()),dict(cats))
