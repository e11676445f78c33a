import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); B=d["config"]["batch_per_launch"]
    print(f, "fps %.0f"%d["value"], "lat", round(d["latency_ms_single_frame"],4), "frac_pipelined", round(d["frame_roofline"]["frac_pipelined"],4))
    tot=0
    for k,v in d["kernel_ms"].items():
        print("   %-18s %8.2f us/launch %6.2f us/frame"%(k, v*1e3, v*1e3/B)); tot+=v*1e3/B
    print("   total us/frame %.1f"%tot)
