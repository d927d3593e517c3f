export PMC_SETS=valu
for f in 0 $((1<<26)) $((1<<24)) $(((1<<24)|(1<<25)|(1<<26)|(1<<27))); do
  bash tools/pmc_kernels.sh floor_$f floor $f > gpurun_out/pmc_floor_$f.log 2>&1 || echo "failed $f"
  python3 -c "
import json
d=json.load(open('gpurun_out/pmc_floor_$f/sq_counters.json'))['kernels']
for k in ('columns_fill_kernel','render_items_kernel'):
    v=d.get(k,{}); w=v.get('SQ_WAVES',1)
    print('flags $f', k, 'waves', w, 'VALU/wave', round(v.get('SQ_INSTS_VALU',0)/w,1), 'SALU/wave', round(v.get('SQ_INSTS_SALU',0)/w,1))
"
done
