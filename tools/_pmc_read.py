import sqlite3, collections, glob, sys
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in 'ab':
    for db in glob.glob(f'/root/repo/gpurun_out/{out}/{f}/*.db'):
        c=sqlite3.connect(db)
        for k,cn,val,dur in c.execute("select kernel_name, counter_name, value, (end-start) from counters_collection"):
            if 'gemm' not in k: continue
            key=k[k.index('gemm')-1:][:34]
            agg[key][cn]+=val
            agg[key]['dur_ns_'+f]+=dur/8
for k,v in agg.items():
    wc=v.get('SQ_WAVE_CYCLES',0); d=v['dur_ns_a']*1e-9
    simd=d*2.1e9*1024
    print(f"{k}: dur {d*1e3:.2f} ms  mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/simd:.3f}  valu/mfma {v['SQ_INSTS_VALU']/(v['SQ_VALU_MFMA_BUSY_CYCLES']/32):.2f}  waves {wc*4/(d*2.1e9)/256:.2f}/CU  wait_any {v['SQ_WAIT_ANY']/wc:.3f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:.3f} active {v['SQ_ACTIVE_INST_ANY']/wc:.3f} valu {v['SQ_ACTIVE_INST_VALU']/wc:.3f} lds {v['SQ_ACTIVE_INST_LDS']/wc:.3f} ldsconf {v['SQ_LDS_BANK_CONFLICT']/max(1,v['SQ_LDS_IDX_ACTIVE']):.3f} coexec {v['SQ_VALU_MFMA_COEXEC_CYCLES']/simd:.3f}")
