cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in variants/base_e56251f.so redux_amd/libredux_hip.so; do
  export REDUX_LIB=$PWD/$lib; tag=$(basename $lib .so)
  timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/wtrace_$tag -o x --output-format csv -- python3 tools/measure_wave.py zipf,1048576,1 > gpurun_out/wtrace_$tag.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/wtrace_$tag/x_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=[r for r in rows if 'k_decode_wave' in r['Kernel_Name']][-1]
i=rows.index(last)
print('$tag')
for r in rows[i-3:i+2]:
    print('  ', r['Kernel_Name'][:50], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
PY
done
