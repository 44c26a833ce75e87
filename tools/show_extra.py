import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)
        for k in ('1GiB','4GiB'):
            print(k, j[k]['encode_GBps'], j[k]['decode_GBps'], j[k]['encode_ms'], j[k]['decode_ms'])
            t=j[k]['timeline_ms [stage begins, enqueued, kernels done, drained] per chunk']
            print(' enc', t['encode'][:10]); print(' dec', t['decode'][:10])
    elif 'gpurun' in l or '==' in l: print(l.strip())
