"""Socket power, shader clock and THROTTLE REASON while the filter kernels run (VERDICT r1 #7c, r2 #2c): samples the amdgpu
hwmon files of THIS process's GPU (power1_input, freq1_input) every ~2 ms around sustained loops of the scan step at the
bench shape — the int8 resident-query form, the int8 half-tile form and the bf16 tier — and reads the SMU's throttler
residency accumulators (amdsmi gpu_metrics: PPT, socket / VR / HBM thermal, prochot) before and after each leg: the share
of the leg in which each limiter held the clock (yams_amd/telemetry.py).  Prints one JSON object
(profiles/r03_power_trace.json is a copy of it)."""
import ctypes, glob, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd import _lib
from yams_amd._lib import SCAN_COSINE


def my_hwmon():
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    hip.hipDeviceGetPCIBusId(buf, 64, 0)
    bus = buf.value.decode().lower()
    for card in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(card)).lower() == bus:
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if hw:
                return hw[0], bus
    return None, bus


class Sampler(threading.Thread):
    def __init__(self, hw):
        super().__init__(daemon=True)
        self.fp = open(os.path.join(hw, "power1_input")); self.ff = open(os.path.join(hw, "freq1_input"))
        self.rows = []; self.stop = False

    def run(self):
        while not self.stop:
            self.fp.seek(0); self.ff.seek(0)
            try:
                self.rows.append((time.perf_counter(), int(self.fp.read()) / 1e6, int(self.ff.read()) / 1e6))
            except ValueError:
                pass
            time.sleep(0.002)


def main():
    n, d, nq, k = int(os.environ.get("ROWS", 12_500_000)), 768, 1024, 100
    hw, bus = my_hwmon()
    acc = Accel(0, torch.cuda.current_stream().cuda_stream)
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
    acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
    acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                           rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda")
    out = {"pci_bus": bus, "hwmon": hw, "shape": f"{n}x{d}, {nq} queries, k={k}", "sample_period_ms": 2,
           "power_cap_W": None, "legs": {}}
    if hw is None:
        out["error"] = "no hwmon directory for this device"; print(json.dumps(out)); return
    try:
        out["power_cap_W"] = int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
    except Exception:
        pass
    smp = Sampler(hw); smp.start()
    time.sleep(0.5)
    from yams_amd import telemetry as ytel
    thr = ytel.Throttle(bus)
    throttle = {}
    marks = {}
    for name, flags in (("int8_resident_query", 0), ("int8_half_tile", _lib.FLAG_WIDE_TILE), ("bf16_single_pass", _lib.FLAG_NO_I8_FILTER)):
        for _ in range(3):
            acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=flags, want_diag=False)
        acc.enable_timing(True)
        snap_a = thr.snapshot()
        t0 = time.perf_counter(); steps = 0
        while time.perf_counter() - t0 < 2.5:
            acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=flags, want_diag=False)
            steps += 1
        t1 = time.perf_counter()
        throttle[name] = thr.between(snap_a, thr.snapshot()) if snap_a else {"error": thr.error}
        fms, fn_ = acc.kernel_ms("scan_filter")
        acc.enable_timing(False)
        marks[name] = (t0, t1, steps, fms)
        time.sleep(0.7)
    smp.stop = True; smp.join()
    idle = [p for (t, p, f) in smp.rows if t < marks["int8_resident_query"][0] - 0.1 and t > smp.rows[0][0] + 0.1]
    out["idle_power_W"] = sum(idle) / max(1, len(idle))
    for name, (t0, t1, steps, fms) in marks.items():
        # skip the first 0.5 s (ramp) of each leg
        rows = [(p, f) for (t, p, f) in smp.rows if t0 + 0.5 <= t <= t1]
        ps = sorted(p for p, _ in rows); fs = sorted(f for _, f in rows)
        q = lambda a, x: a[min(len(a) - 1, int(x * len(a)))] if a else None
        out["legs"][name] = {"steps": steps, "ms_per_step": (t1 - t0) / steps * 1e3, "filter_launch_ms": fms, "samples": len(rows),
                             "power_W": {"mean": sum(ps) / max(1, len(ps)), "p10": q(ps, 0.1), "p50": q(ps, 0.5), "p90": q(ps, 0.9), "max": ps[-1] if ps else None},
                             "sclk_MHz": {"mean": sum(fs) / max(1, len(fs)), "p10": q(fs, 0.1), "p50": q(fs, 0.5), "p90": q(fs, 0.9), "min": fs[0] if fs else None},
                             "throttle": throttle.get(name)}
    print(json.dumps(out))


main()
