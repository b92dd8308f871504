bash tools/pmc_job.sh scan_fwd,scan_bwd,conv,norm,proj,hbm_copy,frontend
ls gpurun_out/pmc | head
python -c "
import json
d=json.load(open('gpurun_out/pmc/pmc_traffic.json'))
for k,v in d.items():
    if not k.endswith('_detail') and k!='_method': print(k, v)
"
