"""Clock, power and THROTTLE-REASON telemetry of the GPU a measurement runs on (plumbing for bench.py and
scripts/power_trace.py; nothing here is on the product path).

Two sources, both optional (every reader returns None when its source is missing):
  * the amdgpu hwmon files of the device (power1_input, freq1_input, power1_cap): sampled every ~2 ms by a thread;
  * the SMU's accumulated throttler residencies through amdsmi (gpu_metrics v1.6+: accumulation_counter,
    ppt / socket-thermal / VR-thermal / HBM-thermal / prochot residency accumulators, and the per-XCC
    "gfx clock below host limit" power / thermal / low-utilisation accumulators of metrics 1.8).  The
    share of a measurement window in which a limiter was active is
        (acc(after) - acc(before)) / (accumulation_counter(after) - accumulation_counter(before))
    (amdsmi.h: "PVIOL % = (PptResidencyAcc(B) - PptResidencyAcc(A)) * 100 / (AccumulationCounter(B) - ...)").
"""
from __future__ import annotations

import ctypes
import glob
import os
import sys
import threading
import time


def hip_pci_bus(device: int = 0) -> str | None:
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, device) != 0:
            return None
        return buf.value.decode().lower()
    except Exception:
        return None


def hwmon_dir(bus: str | None) -> str | None:
    if not bus:
        return None
    for card in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(card)).lower() == bus:
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if hw:
                return hw[0]
    return None


class HwmonSampler(threading.Thread):
    """(time, watts, shader MHz) every ~2 ms while running."""

    def __init__(self, hw: str, period_s: float = 0.002):
        super().__init__(daemon=True)
        self.fp = open(os.path.join(hw, "power1_input"))
        self.ff = open(os.path.join(hw, "freq1_input"))
        self.rows, self.stop, self.period = [], False, period_s

    def run(self):
        while not self.stop:
            self.fp.seek(0); self.ff.seek(0)
            try:
                self.rows.append((time.perf_counter(), int(self.fp.read()) / 1e6, int(self.ff.read()) / 1e6))
            except ValueError:
                pass
            time.sleep(self.period)

    def window(self, t0: float, t1: float):
        rows = [(p, f) for (t, p, f) in self.rows if t0 <= t <= t1]
        if not rows:
            return None
        ps = sorted(p for p, _ in rows); fs = sorted(f for _, f in rows)
        q = lambda a, x: a[min(len(a) - 1, int(x * len(a)))]
        return {"samples": len(rows),
                "power_W": {"mean": sum(ps) / len(ps), "p10": q(ps, 0.1), "p50": q(ps, 0.5), "p90": q(ps, 0.9), "max": ps[-1]},
                "sclk_MHz": {"mean": sum(fs) / len(fs), "p10": q(fs, 0.1), "p50": q(fs, 0.5), "p90": q(fs, 0.9), "min": fs[0]}}


_ACC_FIELDS = ("prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
               "hbm_thm_residency_acc")
_XCP_FIELDS = ("gfx_below_host_limit_ppt_acc", "gfx_below_host_limit_thm_acc", "gfx_low_utilization_acc",
               "gfx_below_host_limit_total_acc")


class Throttle:
    """Throttler residencies of one GPU from the SMU (amdsmi).  usage: a = t.snapshot(); ...; t.between(a, t.snapshot())."""

    def __init__(self, bus: str | None):
        self.h = None
        self.error = None
        try:
            if "/opt/rocm/share/amd_smi" not in sys.path:
                sys.path.append("/opt/rocm/share/amd_smi")
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            for h in amdsmi.amdsmi_get_processor_handles():
                try:
                    bdf = amdsmi.amdsmi_get_gpu_device_bdf(h).lower()
                except Exception:
                    bdf = None
                if bus is None or bdf == bus:
                    self.h = h
                    break
            if self.h is None:
                self.error = f"no amdsmi processor with bdf {bus}"
        except Exception as e:       # noqa: BLE001 - telemetry is optional
            self.error = repr(e)

    def snapshot(self):
        if self.h is None:
            return None
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        except Exception as e:       # noqa: BLE001
            self.error = repr(e)
            return None
        snap = {"t": time.perf_counter(), "acc": m.get("accumulation_counter")}
        for f in _ACC_FIELDS:
            snap[f] = m.get(f)
        # metrics 1.8: per-partition / per-XCC accumulators of "gfx clock below the host limit", by cause
        # (amdsmi returns them as 2-D lists [partition][xcc] under "xcp_stats.<field>", "N/A" where unsupported)
        for f in _XCP_FIELDS:
            tab = m.get("xcp_stats." + f)
            vals = []
            if isinstance(tab, list):
                for part in tab:
                    if isinstance(part, list):
                        vals += [x for x in part if isinstance(x, int) and 0 <= x < (1 << 63)]
            snap[f] = vals or None
        for f in ("temperature_hotspot", "temperature_mem", "temperature_vrsoc", "current_socket_power", "average_gfx_activity",
                  "average_umc_activity", "throttle_status", "indep_throttle_status", "current_uclk", "gfxclk_lock_status"):
            if f in m:
                snap[f] = m.get(f)
        g = m.get("current_gfxclks")
        if isinstance(g, list):
            gg = [x for x in g if isinstance(x, int) and 0 < x < 60000]
            snap["current_gfxclks_MHz"] = gg or None
        return snap

    def violation_status(self):
        """amdsmi's own violation report (percentages since ITS previous reading, and the 'active now' bits)."""
        if self.h is None:
            return None
        try:
            v = self.smi.amdsmi_get_violation_status(self.h)
        except Exception as e:       # noqa: BLE001
            return {"error": repr(e)}
        keep = {}
        for k, x in v.items():
            if k.startswith(("per_", "active_")):
                if isinstance(x, list):      # [partition][xcc]: keep the supported entries of the first partition
                    flat = [y for part in x if isinstance(part, list) for y in part if isinstance(y, (int, bool)) and not isinstance(y, str)]
                    flat = [int(y) for y in flat if int(y) < (1 << 62)]
                    keep[k] = flat[:8] if flat else None
                else:
                    keep[k] = x
        return keep

    @staticmethod
    def _ok(v):
        return isinstance(v, int) and 0 <= v < (1 << 63)

    def between(self, a, b):
        """Share of [a, b] in which each limiter was active, and the state at b."""
        if not a or not b or not self._ok(a.get("acc")) or not self._ok(b.get("acc")) or b["acc"] <= a["acc"]:
            return {"error": self.error or "no accumulation counter"} if (a is None or b is None) else \
                   {"error": "accumulation counter did not advance", "state": {k: b.get(k) for k in b if k not in ("t",)}}
        d = b["acc"] - a["acc"]
        out = {"window_s": b["t"] - a["t"], "accumulation_cycles": d, "source": "amdsmi gpu_metrics residency accumulators"}
        share = {}
        for f in _ACC_FIELDS:
            if self._ok(a.get(f)) and self._ok(b.get(f)):
                share[f.replace("_residency_acc", "")] = (b[f] - a[f]) / d
        for f in _XCP_FIELDS:
            va, vb = a.get(f), b.get(f)
            if isinstance(va, list) and isinstance(vb, list) and len(va) == len(vb) and va:
                per = [(y - x) / d for x, y in zip(va, vb)]
                share[f.replace("_acc", "") + "_per_xcc"] = {"mean": sum(per) / len(per), "max": max(per), "min": min(per)}
        out["active_share"] = share
        out["amdsmi_violation_status_at_end"] = self.violation_status()
        out["at_end"] = {k: b.get(k) for k in ("temperature_hotspot", "temperature_mem", "temperature_vrsoc", "current_socket_power",
                                               "average_gfx_activity", "average_umc_activity", "throttle_status",
                                               "indep_throttle_status", "current_uclk", "current_gfxclks_MHz") if k in b}
        # the limiter: the cause with the largest share, if any was active at all
        flat = {k: (v["mean"] if isinstance(v, dict) else v) for k, v in share.items()}
        if flat:
            k = max(flat, key=flat.get)
            out["limiter"] = k if flat[k] > 0.01 else "none active (no throttler residency accumulated in the window)"
        return out
