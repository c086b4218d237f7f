for t in 16 32 64 128 256; do
  echo -n "DMI_HOST_THREADS=$t  "
  DMI_HOST_THREADS=$t python3 -c "
import sys, time; sys.path.insert(0,'.')
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
meshes = synth.batch_meshes(1024)
w = dmi.meshes_prepare(meshes[:16]); dmi.jobs_encode(w); [j.close() for j in w]
ts=[]
for _ in range(3):
    t0=time.perf_counter(); jobs = dmi.meshes_prepare(meshes); ts.append(time.perf_counter()-t0)
    t0=time.perf_counter(); dmi.jobs_encode_raw(jobs).free(); e=time.perf_counter()-t0
    [j.close() for j in jobs]
print('prepare_s', [round(x,3) for x in ts], 'encode_s', round(e,4))
"
done
