"""Runs scripts/ubench/power_mix (built on the CPU side) for each instruction class while sampling rocm-smi, and prints
package power, sclk and nanojoules per wave-instruction above the `idle` run (test tooling; results under profiles/)."""
import os
import re
import subprocess
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "power_mix")


def run(kind, waves, secs=2.5):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
            pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
            sc = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", o)
            if pw:
                samples.append((float(pw.group(1)), int(sc.group(1)) if sc else 0))

    th = threading.Thread(target=sampler)
    th.start()
    out = subprocess.run([BIN, kind, str(waves), str(secs)], capture_output=True, text=True).stdout
    stop[0] = True
    th.join()
    s = samples[2:-1] or samples
    pw = sum(a for a, _ in s) / max(len(s), 1)
    sc = sum(b for _, b in s) / max(len(s), 1)
    rate = float(re.search(r"wave-instr/s ([\d.e+]+)", out).group(1))
    return pw, sc, rate


idle_pw, _, _ = run("idle", 4, 1.5)
print(f"idle (s_sleep loops on every CU): {idle_pw:.0f} W")
for kind in sys.argv[1:] or ["mfma16", "mfma32", "fma", "exp", "dsr", "mix"]:
    for waves in (4, 8):
        pw, sc, rate = run(kind, waves)
        per = 8 if kind != "mix" else 8       # class instructions per trip (mix: per MFMA slot = 1 MFMA + 1.6 fma + 0.75 exp + 0.4 dsr)
        nj = (pw - idle_pw) / rate * 1e9
        cyc = sc * 1e6 / (rate / (256 * 4))    # cycles per wave-instruction per SIMD (all waves of the SIMD together)
        print(f"{kind:7s} waves/SIMD {waves // 4}: {pw:6.0f} W  sclk {sc:5.0f} MHz  {rate:.3e} wave-instr/s  "
              f"{nj:7.2f} nJ per wave-instr above idle  ({cyc:5.1f} SIMD cycles per instr)", flush=True)
