import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in glob.glob(f'gpurun_out/pmc_{tag}/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void phx::','').replace('phx::','')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        ndisp[k].add(r['Dispatch_Id'])
dur = collections.defaultdict(float)
f = glob.glob(f'gpurun_out/pmc_{tag}/sq1/*/*_kernel_trace.csv')[0]
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void phx::','').replace('phx::','')
    dur[k] += (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for k in agg:
    if not k.startswith('k_'): continue
    a = agg[k]
    print(f"== {k}: {len(ndisp[k])} dispatches, {dur[k]:.2f} ms (under pmc sq1)")
    for c in sorted(a): print(f"   {c:32s} {a[c]:.4g}")
    if 'SQ_WAVE_CYCLES' in a and a['SQ_WAVE_CYCLES']:
        wc = a['SQ_WAVE_CYCLES']
        print(f"   -> wait_any/wave_cycles {a['SQ_WAIT_ANY']/wc:.2f}  wait_inst/wc {a['SQ_WAIT_INST_ANY']/wc:.2f} active_any/wc {a['SQ_ACTIVE_INST_ANY']/wc:.2f} active_valu/wc {a['SQ_ACTIVE_INST_VALU']/wc:.2f}")
        print(f"   -> VALU insts/wave {a['SQ_INSTS_VALU']/a['SQ_WAVES']:.0f}  busy_cycles {a['SQ_BUSY_CYCLES']:.3g}")
    if 'SQ_THREAD_CYCLES_VALU' in a and 'SQ_ACTIVE_INST_VALU' in a:
        print(f"   -> lane utilisation (thread_cycles/(active_valu*64)) ~ {a['SQ_THREAD_CYCLES_VALU']/(a['SQ_ACTIVE_INST_VALU']*4*64):.2f} (if ACTIVE in quad-cycles)")
    if 'TCC_HIT' in a: print(f"   -> L2 hit rate {a['TCC_HIT']/(a['TCC_HIT']+a['TCC_MISS']):.3f}")
    if 'TCP_TOTAL_CACHE_ACCESSES' in a: print(f"   -> L1 miss->L2 read req / L1 accesses {a['TCP_TCC_READ_REQ']/a['TCP_TOTAL_CACHE_ACCESSES']:.3f}")
    if 'FETCH_SIZE' in a: print(f"   -> FETCH_SIZE {a['FETCH_SIZE']/1e6:.2f} GB (KB units; x2 for wide reads per guide)  WRITE_SIZE {a.get('WRITE_SIZE',0)/1e6:.2f} GB")
