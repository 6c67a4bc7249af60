"""Shader clock and socket power WHILE a command runs (amdsmi, 20 ms period):  python tools/clock_during.py <command ...>
Used to price tools/mul_rate_probe.hip's "cycles" (it converts time at an ASSUMED 1.96 GHz) at the clock the chip really held."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rapidsnark_old_amd  # noqa: F401
from rapidsnark_old_amd.telemetry import ClockSampler

with ClockSampler(0, period_s=0.02) as cs:
    t0 = time.time()
    out = subprocess.run(sys.argv[1:], capture_output=True, text=True)
    dt = time.time() - t0
print(out.stdout, end="")
mhz = cs.mhz
print("clock while `%s` ran (%.1f s, %d samples): mean %.0f MHz, min %.0f, max %.0f; samples above 1500 MHz (the loaded phases): mean %.0f MHz; power mean %s W"
      % (" ".join(sys.argv[1:]), dt, len(mhz), sum(mhz) / max(1, len(mhz)), min(mhz or [0]), max(mhz or [0]),
         (lambda hi: sum(hi) / max(1, len(hi)))([m for m in mhz if m > 1500]), cs.summary()["power_w"]))
