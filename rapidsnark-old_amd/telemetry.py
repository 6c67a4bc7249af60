"""Shader clock and socket power of the GPU WHILE a measurement runs (bench.py's issue-bound roofline is priced at the clock the
chip actually held: the proof path keeps an MI355X near its power cap and ~13 % under its 2.4 GHz maximum, DESIGN.md
section 6.3).  Sampled from a background thread through the amdsmi Python binding of the ROCm image (no subprocess in the
timed region); `rocm-smi --json` is the fallback for a single sample.  Measurement plumbing only: nothing here touches the
proof path."""
import threading
import time


def _amdsmi_handle(device):
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    return amdsmi, hs[device if device < len(hs) else 0]


def _read_once(amdsmi, h):
    """-> (gfx clock in MHz | None, socket power in W | None)"""
    mhz = watts = None
    try:
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        per_xcd = [v for v in (m.get("current_gfxclks") or []) if isinstance(v, (int, float)) and 0 < v < 60000]
        if per_xcd:
            mhz = sum(per_xcd) / len(per_xcd)
        elif isinstance(m.get("current_gfxclk"), (int, float)) and 0 < m["current_gfxclk"] < 60000:
            mhz = float(m["current_gfxclk"])
        p = m.get("current_socket_power")
        if isinstance(p, (int, float)) and 0 < p < 60000:
            watts = float(p)
    except Exception:      # noqa: BLE001  (field names differ between amdsmi releases)
        pass
    if mhz is None:
        try:
            c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
            v = c.get("clk", c.get("cur_clk"))
            if isinstance(v, (int, float)) and 0 < v < 60000:
                mhz = float(v)
        except Exception:  # noqa: BLE001
            pass
    if watts is None:
        try:
            p = amdsmi.amdsmi_get_power_info(h)
            v = p.get("current_socket_power", p.get("average_socket_power"))
            if isinstance(v, (int, float)) and 0 < v < 60000:
                watts = float(v)
        except Exception:  # noqa: BLE001
            pass
    return mhz, watts


class ClockSampler:
    """with ClockSampler(device) as cs: ... ; cs.summary() -> {"clock_ghz", "power_w", "samples", "source"} (None values when
    the box offers no telemetry)."""

    def __init__(self, device=0, period_s=0.05, enabled=True):
        self.device, self.period_s = device, period_s
        self.mhz, self.watts = [], []
        self._stop = threading.Event()
        self._th = None
        self.source = None
        self._smi = self._h = None
        if not enabled:
            return
        try:
            self._smi, self._h = _amdsmi_handle(device)
            self.source = "amdsmi (gpu_metrics current_gfxclks / clock_info GFX), sampled every %d ms during the timed region" % int(period_s * 1e3)
        except Exception:  # noqa: BLE001
            self._smi = self._h = None

    def _loop(self):
        while not self._stop.is_set():
            mhz, watts = _read_once(self._smi, self._h)
            if mhz:
                self.mhz.append(mhz)
            if watts:
                self.watts.append(watts)
            self._stop.wait(self.period_s)

    def __enter__(self):
        if self._smi is not None:
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join()
        return False

    def summary(self):
        def mean(xs):
            return sum(xs) / len(xs) if xs else None
        ghz = mean(self.mhz)
        return {"clock_ghz": round(ghz / 1e3, 4) if ghz else None, "power_w": round(mean(self.watts), 1) if self.watts else None,
                "samples": len(self.mhz), "source": self.source or "no telemetry on this box"}
