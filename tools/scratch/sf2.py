import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import siggen, sdr_server_amd as xl
FS=2016000
x = siggen.xs_u8(1, 262144)
for rate in (5, 1, 5, 1):
    taps = xl.create_low_pass_filter(1.0, FS, 24000, 48000 // rate)[1]
    for variant in ("native","optimized"):
        f = xl.XlatingFilter(42, taps, -12000, FS, 262144)
        for _ in range(20): f.process(variant, "cu8", "cf32", x)
        ts=[]
        for _ in range(200):
            t0=time.perf_counter(); f.process(variant, "cu8", "cf32", x); ts.append(time.perf_counter()-t0)
        ts.sort()
        print(variant, len(taps), "mean %.1f med %.1f p10 %.1f p90 %.1f max %.1f us" % (sum(ts)/len(ts)*1e6, ts[100]*1e6, ts[20]*1e6, ts[180]*1e6, ts[-1]*1e6))
        f.close()
