import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), [(x["kernel"], round(x["ms_per_launch"]*1e3,1)) for x in d["kernels"]])
