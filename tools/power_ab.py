"""Dev tool: average package power and shader clock of the training step under two settings of an environment switch —
the energy side of a whole-step A/B ("the step sits on an energy plateau").  A child process runs
`bench.py --steps N` per setting while this process samples `rocm-smi --showpower --showclocks --json` every 0.25 s over
the timed region.  usage: python tools/power_ab.py AVID_TCONV_PARTS 0 7 [steps=400]"""
import json, os, subprocess, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals, steps = sys.argv[1], sys.argv[2:4], int(sys.argv[4]) if len(sys.argv) > 4 else 400


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            p = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
            s = next((v for k, v in card.items() if k.startswith("sclk")), None)
            mhz = float(s.split("(")[1].split("M")[0]) if s and "(" in s else None
            out.append((time.time(), p, mhz))
        except Exception:            # noqa: BLE001
            pass
        time.sleep(0.25)


for rep in range(2):
    for v in vals:
        env = dict(os.environ, **{var: v})
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out), daemon=True)
        p = subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--steps", str(steps), "--warmup", "30", "--no-extra",
                              "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        time.sleep(12.0)             # import, model build, placement probe, warm-up
        th.start()
        line = p.communicate()[0].strip().splitlines()[-1]
        stop.set()
        th.join()
        d = json.loads(line)
        busy = [(pw, mhz) for _, pw, mhz in out if pw and pw > 500]       # samples inside the timed region
        pw = sum(b[0] for b in busy) / max(len(busy), 1)
        ck = [b[1] for b in busy if b[1]]
        print(f"{var}={v}: {d['ms_per_step']:.3f} ms/step, in-bench clock {d['roofline']['shader_clock_ghz']} GHz | rocm-smi over {len(busy)} samples: "
              f"{pw:.0f} W, sclk {sum(ck) / max(len(ck), 1):.0f} MHz | energy per step {pw * d['ms_per_step'] * 1e-3:.2f} J")
