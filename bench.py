#!/usr/bin/env python3
"""bench.py -- proofs/s of the Groth16 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--log-domain 24] [--mix U|C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one proof of a seeded synthetic circuit (ultragroth_amd/synth.py, shapes of SURVEY.md section 8d): the whole hot
path S1-S13 of src/groth16.cpp:48-203 -- the five MSMs and the H-polynomial block on the device, blinding and JSON on the
host -- on a CREATED prover with the witness ALREADY RESIDENT IN HBM when the timed region starts, K steps strictly one
after the other from one host thread: `value` = K / their wall time, `ms_per_step` its inverse, and `msm_ms_per_proof` /
`fft_ms_per_proof` are the device times of the MSM and FFT parts of exactly those K steps. Default workload: configs[2] of
BASELINE.json, the 2^24-constraint circuit with full G1+G2 MSMs that the 10x target is quoted on.
Beside it, outside that region (N = 1): `prove_call_ms_per_step` = the same K proofs through `groth16_prover_prove` with the
.wtns in HOST memory, one call after the other (SURVEY.md section 8(d)'s "ms/proof": parse + PCIe copy of the witness, 512 MiB
at 2^24, + device + host; `witness_upload_ms_per_proof` is the copy's share), and `pipelined_proofs_per_s` = the K calls issued
from `--host-threads` threads on the one prover object (the reference's prover keeps no per-proof state, so its callers may;
here the witness of the waiting call is copied while the kernels of the running one execute). `create_s` = zkey upload,
conversion and window tables.

With N > 1 ranks the base points of every section are sharded by contiguous range (one process per GPU); the witness is
resident as at N = 1 (each rank its slice; the ranks that run an H-polynomial chain all of it). A step: partial sums of the
five MSMs over the rank's slices, the three NTT chains on ranks 0..2 with their evaluation vectors scattered slice-wise over
RCCL, every rank's slice of h and its H-MSM shard, the 384-byte partial records all-gathered over RCCL and added on every
rank (an EC addition is not an RCCL reduction operator), rank 0 finishes. The same proof is produced at every N ("strong"
scaling of one proof). `comm` records what the live process group is (backend, world size, RCCL version).

Rank 0 prints ONE JSON line. `roofline` is for the bucket-accumulation kernel that takes most of a step, the other large
kernels beside it under `roofline.kernels`; all are measured in this run with HIP events on the launch stream inside the
library. `cpu_baseline` times the CPU oracle (oracle/, OpenMP, mulx/adcx/adox Montgomery) on the benchmarked circuit itself
when that takes about two minutes or less (2^24: yes), else on bounded samples with a fitted extrapolation, marked so. The
oracle is the checker/baseline only; the timed path never touches it. `--check` proves once more with fixed blinding and
compares proof.json byte for byte with the expected proof (oracle/closed_form.py: oracle H polynomial + MSMs in the
exponent, any size); a mismatch makes the exit code 3.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-domain", type=int, default=24)
    ap.add_argument("--mix", default="U", choices=["U", "C"])
    ap.add_argument("--host-threads", type=int, default=2, help="N=1: host threads that issue the K timed steps on the one prover object "
                    "(2: the witness copy of a step runs beside the kernels of the step before; 1: strictly one proof after the other)")
    ap.add_argument("--cpu-sample-log", type=int, default=None, help="log2 domain of the CPU-baseline sample circuit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="one more proof with fixed blinding, compared byte for byte with the expected proof "
                                                         "(oracle H polynomial + MSMs in the exponent; any size); exit code 3 on a mismatch")
    ap.add_argument("--g1-only", action="store_true", help="BASELINE.json configs[1]: G1 MSM + NTT only (B1/B2/C sets at infinity)")
    ap.add_argument("--b-zero", type=float, default=0.0, help="(N = 1) fraction of the signals without a B-side point (B1 / B2 at infinity), as in "
                                                                "real circuits; 0 = the dense benchmark circuits of BASELINE.json")
    ap.add_argument("--overlap", type=int, default=None, choices=[0, 1, 2], nargs="?", const=1,
                    help="ULTRAGROTH_OVERLAP of the K timed steps at N = 1: 0 = the H branch behind the witness products, 1 / 2 = beside them on "
                         "the second stream (the library's deployment switch; the fastest honest form and the default here: 1). The per-kernel "
                         "launch times of `roofline` are then taken from K un-overlapped steps right after the region, same process, same prover")
    ap.add_argument("--no-pmc", action="store_true", help="do not start the two rocprofv3 --pmc child runs that measure `roofline.traffic` "
                                                          "(then the committed summary is read and labelled as such)")
    ap.add_argument("--bare", action="store_true", help="(for the counter child runs) only the warm-up and the K steps: no extra figures, "
                                                        "no CPU baseline, no counter children")
    ap.add_argument("--ultra", action="store_true", help="BASELINE.json configs[4]: UltraGroth two-round prove (single GPU)")
    ap.add_argument("--replicas", action="store_true", help="N > 1: after the sharded steps, also time the OTHER way to use a node -- every "
                    "rank a whole prover of its own, proving its own proofs, no exchange at all -- and report it as the extra key "
                    "`replicated_proofs_per_s` (throughput of independent proofs; `value` stays the one proof sharded over the ranks, which "
                    "is what BASELINE.json's configs[3] and the north star ask for). Every rank then builds the whole circuit.")
    ap.add_argument("--pmc-summary", default=None, help="summary of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload "
                    "(tools/pmc_summary.py, made in the same gpurun call by tools/run_r4_prof.sh): `roofline.traffic` is then what those "
                    "passes counted; without it the newest committed profiles/r*_pmc_summary.json is read and labelled as such")
    return ap.parse_args()


def cpu_share():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (v2 or v1), if any"""
    cores = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                quota, period = open(quota_file).read().split()
            else:
                quota, period = open(quota_file).read().strip(), open(period_file).read().strip()
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(dev, args, log_domain, circuit=None):
    """Oracle (plain C + OpenMP; Montgomery product in mulx / adcx / adox form, oracle/field.h) on the SAME generator.
    A 2^20 sample (2^lo in general) is timed first, with the detected CPU share and with 16 and 32 threads when those are
    fewer (the box may show more cores than its share; the fastest setting is kept: the baseline gets every benefit). If the
    benchmarked size is then expected to take at most ~150 s it is proved DIRECTLY (`extrapolated`: false; `circuit` = the
    (zkey, wtns) the GPU just proved, when the caller still holds it). Otherwise 2^hi is timed too and MSM and FFT parts are
    extrapolated with their own fitted exponents (t ~ N^e: Pippenger is sub-linear, the FFT block N log N), marked so."""
    import math
    import oracle as O
    from ultragroth_amd import synth
    share = cpu_share()

    def timed(log, threads, given=None):
        zk, wt = given if given is not None else synth.build_circuit(dev, log, mix=args.mix, g1_only=args.g1_only)[:2]
        O.lib.ugo_set_num_threads(threads)
        t0 = time.perf_counter()
        _, _, (msm_s, fft_s) = O.groth16_prove(zk, wt, 12345, 67890, want_timings=True)
        return time.perf_counter() - t0, msm_s, fft_s, O.lib.ugo_num_threads()

    cap = args.cpu_sample_log if args.cpu_sample_log is not None else log_domain
    lo = min(cap, 20, log_domain)
    best = None
    for threads in sorted({share, min(share, 16), min(share, 32)}):
        t = timed(lo, threads)
        if best is None or t[0] < best[0]:
            best = t
    dt_lo, msm_lo, fft_lo, cores = best
    one_log = min(lo, 16)                                       # for reference: one thread against all of them
    one = timed(one_log, 1)
    many = timed(one_log, cores)
    direct_guess = dt_lo * 2.0 ** (log_domain - lo)             # (linear: an upper bound, Pippenger is sub-linear)
    samples = "2^%d in %.2f s (MSM %.2f | FFT %.2f)" % (lo, dt_lo, msm_lo, fft_lo)
    if log_domain == lo:
        est, msm_s, fft_s, extrapolated, how = dt_lo, msm_lo, fft_lo, False, "measured at the benchmarked size"
    elif cap >= log_domain and direct_guess <= 150.0:
        est, msm_s, fft_s, _ = timed(log_domain, cores, circuit)
        extrapolated, how = False, "the benchmarked 2^%d circuit itself proved in %.1f s (MSM %.1f | FFT %.1f)" % (log_domain, est, msm_s, fft_s)
    else:
        hi = max(min(cap, log_domain - 1, 22), lo + 1) if lo + 1 <= min(cap, log_domain) else lo
        if hi > lo:
            dt_hi, msm_hi, fft_hi, _ = min(timed(hi, cores), timed(hi, cores))      # the host is shared: the faster of two runs
            samples += ", 2^%d in %.2f s (MSM %.2f | FFT %.2f)" % (hi, dt_hi, msm_hi, fft_hi)
            e_msm = math.log2(msm_hi / msm_lo) / (hi - lo)
            e_fft = math.log2(fft_hi / fft_lo) / (hi - lo)
        else:
            dt_hi, msm_hi, fft_hi, e_msm, e_fft = dt_lo, msm_lo, fft_lo, 0.9, 1.0
        rest_hi = max(dt_hi - msm_hi - fft_hi, 0.0)            # parsing, blinding, JSON: grows at most linearly
        k = log_domain - hi
        msm_s, fft_s = msm_hi * 2.0 ** (e_msm * k), fft_hi * 2.0 ** (e_fft * k)
        est = msm_s + fft_s + rest_hi * 2.0 ** k
        extrapolated = True
        how = ("EXTRAPOLATED to 2^%d with the fitted exponents t ~ N^e: MSM e = %.3f, FFT e = %.3f -> %.1f s per proof"
               % (log_domain, e_msm, e_fft, est))
    return {
        "value": 1.0 / est, "unit": "proofs/s", "cores": cores, "kind": "port", "extrapolated": extrapolated,
        "seconds_per_proof": est, "msm_s": msm_s, "fft_s": fft_s,
        "cpu": "%s, nproc %d, share %d" % (cpu_model(), os.cpu_count() or 0, share),
        "sample": "oracle (restated rapidsnark-equivalent CPU path: plain C + OpenMP, mulx/adcx/adox Montgomery product, %d threads, one proof at "
                  "a time, inputs in memory) on circuits of the same generator: %s; %s" % (cores, samples, how),
        "single_thread": {"log_domain": one_log, "seconds_1_thread": one[0], "seconds_all_threads": many[0], "threads": cores},
    }


class BoardSampler:
    """Board power and shader clock sampled from sysfs (the amdgpu hwmon node of the device this rank uses, found by its PCI
    address) every 20 ms while the timed region runs: is the board at its power cap under these kernels? Everything here is
    best effort -- a box that does not show the files gives {"available": false}."""

    def __init__(self, torch, index):
        self.files, self.samples, self.thread, self.stop_flag = {}, [], None, threading.Event()
        self.cap_w = None
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            import glob
            for hw in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % addr):
                for key, names in (("power_uw", ("power1_average", "power1_input")), ("sclk_hz", ("freq1_input",)), ("temp_mc", ("temp2_input", "temp1_input"))):
                    for n in names:
                        if key not in self.files and os.path.exists(os.path.join(hw, n)):
                            self.files[key] = os.path.join(hw, n)
                cap = os.path.join(hw, "power1_cap")
                if os.path.exists(cap):
                    self.cap_w = int(open(cap).read()) / 1e6
            self.addr = addr
        except Exception:                             # noqa: BLE001
            self.files = {}

    def _read(self):
        row = {}
        for k, f in self.files.items():
            try:
                row[k] = int(open(f).read())
            except (OSError, ValueError):
                pass
        return row

    def _run(self):
        while not self.stop_flag.is_set():
            self.samples.append(self._read())
            self.stop_flag.wait(0.02)

    def start(self):
        if self.files:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def stop(self):
        if not self.thread:
            return {"available": False}
        self.stop_flag.set()
        self.thread.join()
        out = {"available": True, "samples": len(self.samples), "pci": self.addr, "source": "amdgpu hwmon (sysfs), every 20 ms over the K timed steps",
               "power_cap_w": self.cap_w}
        pw = [r["power_uw"] / 1e6 for r in self.samples if "power_uw" in r]
        ck = [r["sclk_hz"] / 1e6 for r in self.samples if "sclk_hz" in r]
        tp = [r["temp_mc"] / 1e3 for r in self.samples if "temp_mc" in r]
        if pw:
            out.update(power_w_mean=sum(pw) / len(pw), power_w_max=max(pw))
            if self.cap_w:
                out["power_frac_of_cap_mean"] = sum(pw) / len(pw) / self.cap_w
        if ck:
            out.update(sclk_mhz_mean=sum(ck) / len(ck), sclk_mhz_min=min(ck), sclk_mhz_max=max(ck))
        if tp:
            out["temp_c_max"] = max(tp)
        return out


def measure_traffic(args, log_domain):
    """`roofline.traffic` measured in THIS run: two child runs of this very file under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (counters in passes of their own, with --kernel-trace only, as the microarch guide and gpurun prescribe;
    the program goes directly after `--`; the children are fresh processes -- this one, which has used the GPU, only waits for
    them, its prover closed). Each child builds the same circuit and proves it three times (--bare). Corrected per kernel by
    tools/pmc_summary.py. Returns (summary, label) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    base = tempfile.mkdtemp(prefix="ug_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("ULTRAGROTH_OVERLAP", None)
    t0 = time.perf_counter()
    dirs = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--bare", "--steps", "2", "--warmup", "1", "--overlap", "0",
                   "--log-domain", str(log_domain), "--mix", args.mix, "--host-threads", "1", "--b-zero", str(args.b_zero)] + (["--g1-only"] if args.g1_only else [])
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s child failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-200:])
            dirs[counter] = d
        summary = pmc_summary.summarize(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"], log_domain, how="child runs of this bench.py run, one counter each")
        if os.environ.get("UG_BENCH_PMC_DUMP"):         # (evidence for profiles/: the summary this line's traffic figures come from)
            json.dump(summary, open(os.environ["UG_BENCH_PMC_DUMP"], "w"), indent=1)
        return summary, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of bench.py --bare on the same workload "
                         "(%.0f s), per launch, corrected as the microarch guide says (tools/pmc_summary.py)" % (time.perf_counter() - t0))
    except Exception as e:                            # noqa: BLE001 -- the committed summary is the fallback
        return None, "counter child runs failed: %s" % str(e)[:200]
    finally:
        shutil.rmtree(base, ignore_errors=True)


def witness_slice(info, rank, world):
    """Rank k's slice of the witness-indexed sections. Chain k of the H polynomial runs on rank k mod N beside that
    rank's witness MSMs, so the ranks that carry chains get fewer points: the split is the library's
    (ug_groth16_balanced_witness_range, the one ULTRAGROTH_DEVICES uses inside the library), one source for both launch forms."""
    if world == 1 or info["domainSize"] % world:
        return None                              # even split (and no split H polynomial)
    import ultragroth_amd as ug
    return ug.ShardedGroth16Prover.balanced_witness_range(info["nVars"], rank, world)


def padded_h_slices(h_ranges):
    """How the evaluation vectors travel when the ranks' h ranges differ in length (a bucket-class layout gives its chain ranks
    none): a scatter sends equal pieces, so every rank gets the LONGEST range's length from its own first element on, and the
    vectors carry that much padding behind their end. Returns (first element per rank, piece length)."""
    return [h[0] for h in h_ranges], max(max(h[1] - h[0] for h in h_ranges), 1)


def bench_ultra(args, dev, ug, synth, torch, dist, backend, rank, world, local_rank):
    """UltraGroth (BASELINE.json configs[4]). N = 1: the whole ultra_groth_prover_prove call (witness upload, round
    commitment MSM, Keccak challenge, lookup completion, final-round MSMs + H polynomial, blinding, JSON).
    N > 1: the sharded prover -- every rank uploads the witness and commits to its slice of the round set; the 64-byte
    parts are all-gathered and added, rank 0 closes the round and broadcasts the commitment; every rank applies it and
    runs its slices of the final MSMs; the three NTT chains go to ranks 0..2 with their evaluation vectors scattered
    slice-wise, as for Groth16; the 384-byte partial blocks are all-gathered and rank 0 finishes."""
    LOOKUP_LOG = 16                              # SURVEY.md section 8(d) cfg 5: lookup_size 2^16, chunks M/8
    zkey = None
    if dist is None:
        zkey, uwtns, info = synth.build_ultra_circuit(dev, args.log_domain, mix="C", lookup_log=LOOKUP_LOG)
    else:                                        # N > 1: no rank makes (or holds) the whole zkey -- only its slices, below
        info = synth.ultra_info(args.log_domain)
        uwtns = synth.build_ultra_witness(args.log_domain, "C", lookup_log=LOOKUP_LOG)
    workload = ("ultragroth-bn254 2^%d constraints, two rounds, lookup 2^%d, circom-like witness (BASELINE.json configs[4] "
                "shape; step = ultra_groth_prover_prove, one call after the other, .uwtns in host memory: the lookup completion "
                "rewrites the witness every proof, so there is no resident-witness form of this step)" % (args.log_domain, LOOKUP_LOG))
    FIXED = (bytes(range(1, 32)), bytes(range(40, 71)), bytes(range(80, 111)))          # r_k, r, s of --check

    def expected():
        import oracle as O
        zk = zkey if zkey is not None else synth.build_ultra_circuit(dev, args.log_domain, mix="C", lookup_log=LOOKUP_LOG)[0]
        return O.ultra_groth_prove(zk, uwtns, *(int.from_bytes(b, "little") for b in FIXED))

    def line(elapsed, msm_ms, fft_ms, create_s, parallelism, **extra):
        print(json.dumps({
            "metric": "proofs/s", "value": args.steps / elapsed, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 254-bit modular integers)", "data": "synthetic",
            "config": {"workload": workload, "log_domain": args.log_domain, "protocol": "ultragroth", "parallelism": parallelism},
            "msm_ms_per_proof": msm_ms / args.steps, "fft_ms_per_proof": fft_ms / args.steps, "create_s": create_s,
            **extra,
        }))

    if dist is None:                             # (UG_BENCH_FORCE_DIST=1: the sharded path below over a process group of one rank)
        t0 = time.perf_counter()
        prover = ug.UltraGrothProver(zkey)
        create_s = time.perf_counter() - t0
        for _ in range(args.warmup):
            prover.prove(uwtns)
        msm_ms = fft_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            prover.prove(uwtns)
            m, f, _ = prover.last_timings()
            msm_ms += m
            fft_ms += f
        elapsed = time.perf_counter() - t0                 # THE timed region: K whole calls, one after the other
        # the region ran in the library's default configuration (the H branch of the final round beside its witness products:
        # stream times of the two parts stretch and overlap); the MSM | FFT split of the metric comes from K more calls with the
        # branch behind the products
        unoverlapped_ms = None
        if os.environ.get("ULTRAGROTH_OVERLAP", "1") != "0":
            keep = os.environ.get("ULTRAGROTH_OVERLAP")
            os.environ["ULTRAGROTH_OVERLAP"] = "0"
            prover.prove(uwtns)
            msm_ms = fft_ms = 0.0
            tc = time.perf_counter()
            for _ in range(args.steps):
                prover.prove(uwtns)
                m, f, _ = prover.last_timings()
                msm_ms += m
                fft_ms += f
            unoverlapped_ms = 1e3 * (time.perf_counter() - tc) / args.steps
            if keep is None:
                del os.environ["ULTRAGROTH_OVERLAP"]
            else:
                os.environ["ULTRAGROTH_OVERLAP"] = keep
        pipelined_s = None
        if args.host_threads > 1:
            # an extra figure, as on the Groth16 line: the K calls from several host threads on the one prover object (the .uwtns
            # of a waiting call is staged while a proof runs)
            todo = iter(range(args.steps))
            tally = threading.Lock()
            failures = []

            def issue_steps():
                while True:
                    with tally:
                        if failures or next(todo, None) is None:
                            return
                    try:
                        prover.prove(uwtns)
                    except BaseException as e:
                        with tally:
                            failures.append(e)
                        return
            t0 = time.perf_counter()
            helpers = [threading.Thread(target=issue_steps) for _ in range(args.host_threads - 1)]
            for th in helpers:
                th.start()
            issue_steps()
            for th in helpers:
                th.join()
            if failures:
                raise failures[0]
            pipelined_s = time.perf_counter() - t0
        ok = True
        if args.check:
            ug.set_test_blinding(b"".join(FIXED))
            got = prover.prove(uwtns)
            ug.set_test_blinding(b"")
            ok = got == expected()
            workload += " [check: %s]" % ("bit-exact" if ok else "MISMATCH")
        line(elapsed, msm_ms, fft_ms, create_s, "one GPU", pipelined_proofs_per_s=(args.steps / pipelined_s) if pipelined_s else None,
             pipelined_host_threads=max(1, args.host_threads) if pipelined_s else None, unoverlapped_ms_per_step=unoverlapped_ms,
             split_region=("device time of the MSM and FFT parts of K more calls with the H branch behind the witness products (ULTRAGROTH_OVERLAP=0)"
                           if unoverlapped_ms else "device time of the MSM and FFT parts of the K timed calls"),
             value_definition="K / wall time of K ultra_groth_prover_prove calls (src/prover.h), .uwtns in host memory, one after the other; "
                              "library switches: ULTRAGROTH_OVERLAP=%s" % os.environ.get("ULTRAGROTH_OVERLAP", "1 (default)"))
        if not ok:
            sys.exit(3)
        return

    n_dom = info["domainSize"]
    split_h = n_dom % world == 0
    my_chains = [k for k in range(3) if k % world == rank] if split_h else list(range(3))
    # every rank from the header section and ITS slices of the point sections and index lists (ug_ultra_groth_prover_create_sharded_slices;
    # the same generator entered at the slice), the coefficient records only on the ranks that run a chain
    rg = ug.ShardedUltraGrothProver.shard_ranges(info["nVars"], n_dom, info["nC1"], info["nC2"], rank, world)
    header, coefs, slices = synth.build_ultra_circuit_slices(dev, args.log_domain, rg, with_coefs=bool(my_chains))
    t0 = time.perf_counter()
    prover = ug.ShardedUltraGrothProver.from_slices(header, coefs, info["nCoefs"], slices, local_rank, rank, world,
                                                    public_size=82 * (info["nPublic"] - 1) + 4)
    create_s = time.perf_counter() - t0
    del coefs, slices
    cuda = backend == "nccl"
    sl = n_dom // world if split_h else 0
    fulls = {k: torch.empty((n_dom, 32), dtype=torch.uint8, device="cuda") for k in my_chains}
    bufs = torch.empty((3, max(sl, 1), 32), dtype=torch.uint8, device="cuda")

    def to_comm(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        return t.cuda() if cuda else t

    # the small blocks of a proof (64-byte commitment parts, 384-byte partial sums) of all ranks: one collective into one
    # preallocated buffer and one copy back through pinned memory, as on the Groth16 path
    if cuda:
        xmit = {n: torch.empty(n, dtype=torch.uint8, device="cuda") for n in (64, 384)}
        recv = {n: torch.empty(n * world, dtype=torch.uint8, device="cuda") for n in (64, 384)}
        stage = {n: torch.empty(n, dtype=torch.uint8).pin_memory() for n in (64, 384)}
        landed = {n: torch.empty(n * world, dtype=torch.uint8).pin_memory() for n in (64, 384)}

    def gather_all(b):
        n = len(b)
        if cuda and n in xmit:
            stage[n].copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
            xmit[n].copy_(stage[n], non_blocking=True)
            dist.all_gather_into_tensor(recv[n], xmit[n])
            landed[n].copy_(recv[n], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            raw = bytes(landed[n].numpy())
            return [raw[n * q:n * (q + 1)] for q in range(world)]
        mine = to_comm(b)
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        return [bytes(t.cpu().numpy()) for t in allp]

    comm_stream = torch.cuda.Stream() if (split_h and cuda) else None
    chain_order = os.environ.get("UG_BENCH_CHAIN_ORDER", "auto")          # as on the Groth16 path (main())
    chain_first = chain_order == "first" or (chain_order == "auto" and world >= 5)

    def run_chains():
        for k in my_chains:
            prover.hpoly_chain(k, fulls[k].data_ptr())

    def step():
        prover.load_witness(uwtns)
        total = bytes(64)
        for part in gather_all(prover.round_commit()):
            total = ug.ShardedUltraGrothProver.add_records(total, part)
        commitment = to_comm(prover.round_finish(total) if rank == 0 else bytes(64))
        dist.broadcast(commitment, src=0)
        prover.apply_commitment(bytes(commitment.cpu().numpy()))
        if split_h:
            # the final round as on the Groth16 path: witness products queued (A | B1, B2, the gathered final set), the H branch
            # on the library's second stream beside them; a chain rank of a large node runs its chain first
            if chain_first:
                run_chains()
                prover.witness_msm_begin()
            else:
                prover.witness_msm_begin()
                run_chains()
            if comm_stream is not None:
                with torch.cuda.stream(comm_stream):
                    works = [dist.scatter(bufs[k], [fulls[k][q * sl:(q + 1) * sl] for q in range(world)] if rank == k % world else None,
                                          src=k % world, async_op=True) for k in range(3)]
                    for wk in works:
                        wk.wait()
                comm_stream.synchronize()
            else:                               # gloo rehearsal: through host memory
                for k in range(3):
                    src = k % world
                    o = torch.empty(bufs[k].shape, dtype=torch.uint8)
                    dist.scatter(o, [fulls[k][q * sl:(q + 1) * sl].cpu() for q in range(world)] if rank == src else None, src=src)
                    bufs[k].copy_(o)
                torch.cuda.current_stream().synchronize()
            prover.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
            hpart = prover.run_h_msm()
            part = prover.witness_msm_end()[:320] + hpart[320:384]
        else:                                   # the domain does not split evenly: every rank forms h itself
            part = prover.run_witness_msm()
            for k in range(3):
                prover.hpoly_chain(k, fulls[k].data_ptr())
            first, cnt, _ = prover.h_range()
            sl_bufs = [fulls[k][first:first + cnt].contiguous() for k in range(3)]
            torch.cuda.synchronize()
            prover.hpoly_combine(*(b.data_ptr() for b in sl_bufs))
            part = part[:320] + prover.run_h_msm()[320:384]
        acc = None
        for other in gather_all(part):
            acc = other if acc is None else ug.ShardedGroth16Prover.add_partials(acc, other)
        return prover.finish(acc) if rank == 0 else None

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    msm_ms = fft_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        m, f, _ = prover.last_timings()
        msm_ms += m
        fft_ms += f
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    chk = None
    if args.check:
        ug.set_test_blinding(b"".join(FIXED))
        chk = step()
        ug.set_test_blinding(b"")
    ok = True
    dist.barrier()
    dist.destroy_process_group()          # before rank 0's host-side comparison (the CPU oracle proves the circuit): nobody waits in a collective
    if rank == 0:
        if args.check:
            ok = chk == expected()
            workload += " [check: %s]" % ("bit-exact" if ok else "MISMATCH")
        line(float(t.item()), msm_ms, fft_ms, create_s,
             "section-range shard x%d%s" % (world, ", H-poly chains split over ranks" if split_h else ""))
    if not ok:
        sys.exit(3)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    if args.check:
        os.environ["ULTRAGROTH_TEST_HOOKS"] = "1"       # fixed blinding for the comparison (only honoured when set before load)
    import torch
    import ultragroth_amd as ug
    from ultragroth_amd import synth
    if args.bare:
        args.no_pmc = args.no_cpu_baseline = True
    # the K timed steps at N = 1 run in the fastest honest configuration of the library: the H branch beside the witness
    # products (ULTRAGROTH_OVERLAP, read by the library per proof); N > 1 drives the two streams itself (phase calls)
    if world == 1 and args.ultra and args.overlap:
        os.environ["ULTRAGROTH_OVERLAP"] = str(args.overlap)

    # UG_BENCH_BACKEND=gloo + UG_BENCH_ONE_DEVICE=1 rehearse the N > 1 control flow on a one-GPU box
    # (RCCL refuses two ranks on one device); the driver's multi-GPU runs use the defaults: nccl, one GPU per rank.
    backend = os.environ.get("UG_BENCH_BACKEND", "nccl")
    if os.environ.get("UG_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # UG_BENCH_FORCE_DIST=1: take the N > 1 code path with a process group of ONE rank (a one-GPU box can then execute the real
    # RCCL calls of that path -- scatter, all_gather_into_tensor, all_reduce, barrier -- instead of none at all)
    forced = world == 1 and os.environ.get("UG_BENCH_FORCE_DIST") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    dev = ug.Device(local_rank)
    log_domain = args.log_domain
    if args.ultra:
        return bench_ultra(args, dev, ug, synth, torch, dist, backend, rank, world, local_rank)
    zkey = None
    single = dist is None                       # the one-GPU form: the reference's calls on one prover object
    timed_overlap = (1 if args.overlap is None else args.overlap) if single else 0
    if single:
        zkey, wtns, info = synth.build_circuit(dev, log_domain, mix=args.mix, g1_only=args.g1_only, b_zero=args.b_zero)
        zkey_bytes = len(zkey)
        t0 = time.perf_counter()
        prover = ug.Groth16Prover(zkey)          # groth16_prover_create: the reference's own entry point
        create_s = time.perf_counter() - t0
        # cold start: create does not wait for the window tables; the first proof (the reference's call, .wtns in host memory)
        # goes at once, on the classic windows beside the table kernels; the timed region below runs on the finished tables
        first_proof_s = tables_s = None
        if not args.bare:
            prover.prove(wtns)
            first_proof_s = time.perf_counter() - t0
        prover.tables_ready(wait=True)
        tables_s = time.perf_counter() - t0
        if not args.check and args.no_cpu_baseline:
            del zkey
            zkey = None
    else:
        # every rank makes ONLY its slices of the point sections (same generator walk, entered at the slice); the
        # coefficient records only on the ranks that run an H-polynomial chain: no rank holds the whole zkey
        domain = 1 << log_domain
        info = dict(domainSize=domain, nVars=domain - 1, nPublic=1, nCoefs=4 * domain)
        # the layout is the library's (ug_groth16_shard_layout: what ULTRAGROTH_DEVICES uses inside it): base-point ranges with
        # fewer points for the chain ranks, or -- ULTRAGROTH_SHARD=PxB -- bucket classes; a domain that does not split over the
        # ranks keeps the even base-point form (every rank then forms the whole H polynomial itself)
        even = domain % world != 0
        layouts = [ug.ShardedGroth16Prover.shard_layout(info["nVars"], 1, domain, r, world, world if even else 0) for r in range(world)]
        layout = layouts[rank]
        if even:
            rg = ug.ShardedGroth16Prover.shard_ranges(info["nVars"], 1, domain, rank, world, None)
            layout = ug.ShardLayout([rg[0][0], rg[0][1], rg[1][0], rg[1][1], rg[2][0], rg[2][1], 0, 0, 1, rg[0][0], rg[0][1], 7])
        runs_chain = even or bool(layout.chains)
        header, coefs, slices = synth.build_circuit_slices(dev, log_domain, layout.ranges, with_coefs=runs_chain, g1_only=args.g1_only)
        wtns = synth.build_witness(log_domain, args.mix)
        zkey_bytes = len(header) + (len(coefs) if coefs is not None else 0) + sum(len(x) for x in slices)
        t0 = time.perf_counter()
        prover = ug.ShardedGroth16Prover.from_slices(header, coefs, info["nCoefs"], slices, local_rank, rank, world,
                                                     public_size=82 + 4, layout=layout)
        create_s = time.perf_counter() - t0
        del coefs, slices

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the three iFFT/twist/FFT chains of the H polynomial go to ranks 0..2 (k mod N); each source rank scatters
    # the slices of its evaluation vector over RCCL, so every rank holds its slice of all three and forms its slice of h.
    split_h = dist is not None and info["domainSize"] % world == 0
    if split_h:
        n_dom = info["domainSize"]
        # every rank's slice of h (the layouts' h ranges: even in the base-point form; in a bucket-class layout of five ranks or
        # more the chain ranks take none). A scatter sends equal pieces: the longest range's length, from each rank's first
        # element on (the vectors carry that much padding behind their end)
        h_first, sl = padded_h_slices([L.h for L in layouts])
        ev_dev = "cuda"
        fulls = {k: torch.empty((n_dom + sl, 32), dtype=torch.uint8, device=ev_dev) for k in layout.chains}
        bufs = torch.empty((3, sl, 32), dtype=torch.uint8, device=ev_dev)

    def scatter_slices(out, src_full, src):
        """returns a work handle (nccl: the three scatters, from three different roots, are in flight together) or None"""
        if backend == "nccl":
            return dist.scatter(out, [src_full[h_first[r]:h_first[r] + sl] for r in range(world)] if rank == src else None, src=src,
                                async_op=True)
        o = torch.empty(out.shape, dtype=torch.uint8)               # gloo rehearsal: through host memory
        lst = [src_full[h_first[r]:h_first[r] + sl].cpu() for r in range(world)] if rank == src else None
        dist.scatter(o, lst, src=src)
        out.copy_(o)
        return None

    my_chains = list(layout.chains) if split_h else []

    # the 384-byte partial blocks of all ranks: ONE collective into one buffer and one copy back to the host (an EC addition is
    # not an RCCL reduction operator, so the blocks are gathered and added on the host: 5 points per rank)
    if dist is not None and backend == "nccl":
        xmit = torch.empty(384, dtype=torch.uint8, device="cuda")
        recv = torch.empty(384 * world, dtype=torch.uint8, device="cuda")
        stage = torch.empty(384, dtype=torch.uint8).pin_memory()
        landed = torch.empty(384 * world, dtype=torch.uint8).pin_memory()

    def exchange_partials(part):
        if backend == "nccl":
            stage.copy_(torch.frombuffer(bytearray(part), dtype=torch.uint8))
            xmit.copy_(stage, non_blocking=True)
            dist.all_gather_into_tensor(recv, xmit)
            landed.copy_(recv, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            raw = bytes(landed.numpy())
            return [raw[384 * r:384 * (r + 1)] for r in range(world)]
        mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)      # gloo rehearsal
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        return [bytes(t.numpy()) for t in allp]

    def load():
        """the witness into HBM (outside the timed region: BASELINE contract -- inputs resident when the timed region starts)"""
        if single or not split_h:
            prover.load_witness(wtns)
            return
        prover.load_witness_part(wtns, 0)        # this rank's slice of the scalars: all its witness MSMs read
        if my_chains:
            prover.load_witness_part(wtns, 1)    # the rest: only the mat-vec of a chain rank reads it

    def run_chains():
        for k in my_chains:
            prover.hpoly_chain(k, fulls[k].data_ptr())

    # the collectives of a step run on a stream of their own: torch's default stream is the legacy null stream, and nothing of
    # the step should be ordered against it
    comm_stream = torch.cuda.Stream() if (split_h and backend == "nccl") else None
    sequential_phases = os.environ.get("UG_BENCH_PHASES") == "sequential"       # A/B: the blocking phase calls of rounds 1-3
    # Where a chain rank starts its chain (measured, tools/run_r3_order.sh, rank 0 of 8 at 2^24: the stream priority class has no
    # effect on this stack -- a chain queued beside the witness products gets ~40 % of the chip and takes 17 ms instead of 7 --
    # so on a node where most ranks WAIT for the chains, the chain ranks run theirs first, alone, and queue their products
    # behind; with few ranks (long products, every rank busy anyway) the chain runs beside the products, which gains what a
    # memory-bound kernel gains beside an issue-bound one). UG_BENCH_CHAIN_ORDER = first | beside | auto (first from 5 ranks on).
    chain_order = os.environ.get("UG_BENCH_CHAIN_ORDER", "auto")
    chain_first = chain_order == "first" or (chain_order == "auto" and world >= 5)

    def step():
        """one proof from the witness resident in HBM: S1-S13 of src/groth16.cpp:48-203"""
        if single:
            return prover.prove_resident()       # ug_groth16_prover_prove_resident: groth16_prover_prove minus parse and copy
        if split_h and sequential_phases:
            th = None
            if my_chains:                        # the H branch has its own stream inside the library: chains beside the MSMs
                th = threading.Thread(target=run_chains)
                th.start()
            part = prover.run_witness_msm()
            if th is not None:
                th.join()
            works = [scatter_slices(bufs[k], fulls.get(k), k % world) for k in range(3)]
            for wk in works:
                if wk is not None:
                    wk.wait()
            torch.cuda.synchronize()
            prover.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
            part = part[:320] + prover.run_h_msm()[320:384]
        elif split_h:
            # ONE host thread, both streams of the rank busy: the witness products are queued and left to run; the H branch --
            # this rank's chains, the slice exchange, combine, the H product, all on the library's second (high-priority)
            # stream and torch's collectives -- is driven meanwhile, so the ranks that wait for a chain rank's evaluation
            # vectors spend that time on their witness products, and the H product's latency-bound tail runs beside them
            if chain_first:
                run_chains()                     # (returns when this rank's evaluation vectors are complete)
                prover.witness_msm_begin()
            else:
                prover.witness_msm_begin()
                run_chains()
            if comm_stream is not None:
                with torch.cuda.stream(comm_stream):
                    works = [scatter_slices(bufs[k], fulls.get(k), k % world) for k in range(3)]
                    for wk in works:
                        wk.wait()
                comm_stream.synchronize()        # the slices are here; the witness stream is NOT waited for
            else:                                # (gloo rehearsal: through host memory, on torch's current stream)
                for k in range(3):
                    scatter_slices(bufs[k], fulls.get(k), k % world)
                torch.cuda.current_stream().synchronize()
            prover.hpoly_combine(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
            hpart = prover.run_h_msm()
            part = prover.witness_msm_end()[:320] + hpart[320:384]
        else:                                    # the domain does not split evenly: every rank forms h itself
            part = prover.run()
        parts = exchange_partials(part)
        total = parts[0]
        for other in parts[1:]:
            total = prover.add_partials(total, other)
        return prover.finish(total) if rank == 0 else None

    out = None
    load()
    os.environ["ULTRAGROTH_OVERLAP"] = str(timed_overlap)
    for _ in range(args.warmup):
        out = step()
    for which in range(4):
        prover.kernel_stats(which=which, reset=True)      # (switches the per-kernel event pairs on: they are part of the region)
    if os.environ.get("ULTRAGROTH_GRAPH", "0") not in ("", "0") and single:
        for _ in range(2):                                # a recorded launch sequence is recorded outside the region (with the event pairs)
            out = step()
        for which in range(4):
            prover.kernel_stats(which=which, reset=True)
    sampler = BoardSampler(torch, local_rank) if rank == 0 else None
    # ---- THE timed region: K proofs, one after the other, witness resident in HBM ----
    # (the library's device-time accumulators restart when a witness is loaded: with the witness resident they run on, so the
    # K steps' share is the difference across the region)
    m0, f0, _ = prover.last_timings()
    barrier()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    board = sampler.stop() if sampler else None
    m1, f1, _ = prover.last_timings()
    msm_ms, fft_ms = m1 - m0, f1 - f0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kstats_region = [prover.kernel_stats(which=w) for w in range(4)]      # of the K timed steps only (reset after the warm-up)
    kstats, clean_ms, clean_split = kstats_region, None, None
    if timed_overlap:
        # The region ran with the H branch BESIDE the witness products: launches of the two streams share the chip and their
        # event-pair times stretch. The launch durations the roofline is computed from are therefore taken from K more steps with
        # the branch BEHIND the products (one kernel on the chip at a time), right here: same process, same prover, same witness.
        os.environ["ULTRAGROTH_OVERLAP"] = "0"
        step()
        for which in range(4):
            prover.kernel_stats(which=which, reset=True)
        mc0, fc0, _ = prover.last_timings()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        clean_ms = 1e3 * (time.perf_counter() - tc) / args.steps
        mc1, fc1, _ = prover.last_timings()
        clean_split = ((mc1 - mc0) / args.steps, (fc1 - fc0) / args.steps)
        kstats = [prover.kernel_stats(which=w) for w in range(4)]
        os.environ["ULTRAGROTH_OVERLAP"] = str(timed_overlap)

    # ---- extra figures (N = 1), outside the contract's region: the same K proofs through groth16_prover_prove with the .wtns
    # in HOST memory -- SURVEY.md section 8(d)'s "ms/proof": parse + PCIe copy + device + host -- one call after the other, and
    # from `--host-threads` threads on the one prover object (the witness copy of a call runs beside the kernels of the other)
    host_threads = max(1, args.host_threads) if single else 1
    prove_call_ms = upload_ms = pipelined_s = None
    if single and not args.bare:
        prover.prove(wtns)
        t1 = time.perf_counter()
        upload_ms = 0.0
        for _ in range(args.steps):
            out = prover.prove(wtns)
            upload_ms += prover.last_upload_ms()
        prove_call_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        upload_ms /= args.steps
        if host_threads > 1:
            tally = threading.Lock()
            todo = iter(range(args.steps))
            failures = []

            def issue_steps():
                while True:
                    with tally:
                        if failures or next(todo, None) is None:
                            return
                    try:
                        prover.prove(wtns)
                    except BaseException as e:      # a helper thread must not lose a step silently: the run fails below
                        with tally:
                            failures.append(e)
                        return
            t1 = time.perf_counter()
            helpers = [threading.Thread(target=issue_steps) for _ in range(host_threads - 1)]
            for th in helpers:
                th.start()
            issue_steps()
            for th in helpers:
                th.join()
            if failures:
                raise failures[0]
            pipelined_s = time.perf_counter() - t1

    chk = None
    if args.check:                              # one more step on EVERY rank (it contains collectives), fixed blinding
        ug.set_test_blinding(bytes(range(1, 32)) + bytes(range(31, 62)))
        chk = step()
        if single:                              # ... and once more through the reference's entry point, witness in host memory
            ug.set_test_blinding(bytes(range(1, 32)) + bytes(range(31, 62)))
            if prover.prove(wtns) != chk:
                chk = ("groth16_prover_prove differs from the phase calls", "")
        ug.set_test_blinding(b"")

    # --replicas: the node used the other way round -- N independent provers, each proving whole proofs on its own GPU (what a
    # batch of unrelated proofs wants: no exchange, linear by construction). An extra figure; a failure on any rank costs the
    # figure, not the line (the failure is made collective, below).
    replicated = None
    if dist is not None and args.replicas:
        # Every collective of this block sits OUTSIDE the per-rank try: a rank that fails (out of memory building the whole circuit,
        # say) still takes part in the agreement below, so that no rank waits in a barrier for one that has left. The ranks agree on
        # an ok flag (MIN) before the timed loop and after it; if any rank failed, all skip the rest and report the error.
        flag_dev = "cuda" if backend == "nccl" else "cpu"

        def all_ok(mine):
            f = torch.tensor([1 if mine else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return bool(f.item())

        ms_sharded = 1e3 * elapsed / args.steps
        own, err, dt_all = None, None, None
        try:
            prover.close()
            os.environ["ULTRAGROTH_DEVICE"] = str(local_rank)       # (the reference's create call takes its device from here)
            full_zkey, full_wtns, _ = synth.build_circuit(dev, log_domain, mix=args.mix, g1_only=args.g1_only)
            own = ug.Groth16Prover(full_zkey)
            del full_zkey
            own.load_witness(full_wtns)
            own.prove_resident()
        except Exception as e:                          # noqa: BLE001 -- reported, not fatal
            err = str(e)[:300]
        if all_ok(err is None):
            barrier()
            t1 = time.perf_counter()
            try:
                for _ in range(args.steps):
                    own.prove_resident()
            except Exception as e:                      # noqa: BLE001
                err = str(e)[:300]
            barrier()
            dt = time.perf_counter() - t1
            if all_ok(err is None):
                tt = torch.tensor([dt], dtype=torch.float64, device=flag_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_all = float(tt.item())
        if own is not None:
            try:
                own.close()
            except Exception:                           # noqa: BLE001
                pass
        if dt_all is not None:
            replicated = {"proofs_per_s": world * args.steps / dt_all, "ms_per_proof_per_rank": 1e3 * dt_all / args.steps,
                          "note": "every rank a whole prover of its own proving its own proofs (no exchange); against %.2f ms per proof "
                                  "for ONE proof sharded over the %d ranks" % (ms_sharded, world)}
        else:
            replicated = {"proofs_per_s": None, "error": err or "another rank failed"}

    # The process group ends HERE, before rank 0 assembles the line: what follows on rank 0 (the --check comparison, which
    # synthesises the whole circuit once more; at N = 1 the CPU baseline) is host work of tens of seconds, and the other ranks
    # must not sit in a collective -- spinning on their GPUs -- while it runs: they are done and exit.
    comm = None
    if dist is not None:
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            ver = None
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
                "devices_visible": torch.cuda.device_count()}
        dist.barrier()
        dist.destroy_process_group()

    rc = 0
    if rank == 0:
        (acc_ms, launches, entries), (g2_ms, g2_launches, g2_entries), (ntt_ms, ntt_launches, ntt_points) = kstats[:3]
        grp_ms, grp_launches, grp_entries = kstats[3]                          # A | B1 | C in one launch (three products per entry)
        n_local = (layouts[0].witness[1] - layouts[0].witness[0]) if not single else info["nVars"]
        # Algorithmic bytes per launch (restated in DESIGN.md): G1 accumulation 96 B per point of
        # the slice (64 B affine base + 32 B scalar, each read once), G2 160 B per point, one NTT pass 64 B per point
        g1_bytes, g2_bytes, ntt_bytes = 96.0 * n_local, 160.0 * n_local, 64.0 * info["domainSize"]
        # the group launch reads three 64-byte bases and one 32-byte scalar per point: 224 B per point ([A | C], 160 B, when the
        # prover keeps B1 / B2 compacted over the signals with a real B point: --b-zero from a quarter on)
        grp_members = 2 if (single and args.b_zero >= 0.25 and os.environ.get("ULTRAGROTH_SPARSE_B", "1") != "0") else 3
        grp_bytes = (32.0 + 64.0 * grp_members) * n_local

        def gbs(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        def mads(per_unit, units, n, ms):
            return (per_unit * units / max(n, 1) / (ms * 1e-3) / 1e12) if ms > 0 else 0.0
        # One entry per large kernel, all measured in THIS run with HIP events on the launch stream inside the library.
        # Algorithmic bytes per launch: each base and each scalar read once.
        # issue_bound: the roof that actually binds -- v_mad_u64_u32 issue, 29 T mad/s measured (tools/ubench_int.hip); one G1
        # mixed addition = 1467 mads, G2 4470
        def entry(name, nbytes, ms, n_launch, units, mad_per_unit=None, **extra):
            e = {"bound": "hbm", "kernel": name, "achieved": gbs(nbytes, ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": gbs(nbytes, ms) / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                 "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "launches": n_launch,
                 "ms_per_step": ms * n_launch / max(args.steps, 1)}
            if mad_per_unit:
                e["issue_bound"] = {"unit": "T mad/s", "peak": 29.0, "achieved": mads(mad_per_unit, units, n_launch, ms),
                                    "frac": mads(mad_per_unit, units, n_launch, ms) / 29.0}
                # `achieved` / `peak` / `frac` stay the HBM figures the north star and the run contract ask for (algorithmic bytes
                # against 8 TB/s); `bound` NAMES the roof that is the nearer one for this kernel -- the larger of the two
                # fractions -- and `bound_frac` is that fraction
                issue = e["issue_bound"]["frac"] > e["frac"]
                e["bound"] = "valu-issue" if issue else "hbm"
                e["bound_frac"] = e["issue_bound"]["frac"] if issue else e["frac"]
                e["limiter"] = "valu-issue (v_mad_u64_u32)" if issue else "hbm"
            e.update(extra)
            return e
        # multiply-adds of the NTT launches per point and pass: a butterfly (162) per two points and stage, and the products folded
        # into first / last passes (three twists, a o b, a o b - c and its conversion): 162 * (6 * logn / 2 + 6) per point of the
        # domain, over the 18 pass-transforms of a proof
        ntt_passes = max(1, -(-(log_domain - 10) // 8) + 1) if log_domain > 10 else 1
        ntt_mads = 162.0 * (3.0 * log_domain + 6.0) / (6.0 * ntt_passes)
        kern = [entry("segment_accumulate_kernel<G1Cfg>", g1_bytes, acc_ms, launches, entries, 1467.0),
                entry("segment_accumulate_kernel<G2Cfg>", g2_bytes, g2_ms, g2_launches, g2_entries, 4470.0),
                entry("segment_accumulate_group_kernel<%d>" % grp_members, grp_bytes, grp_ms, grp_launches, grp_entries, 1467.0, products_per_launch=grp_members),
                entry("ntt_pass_kernel", ntt_bytes * (ntt_points / max(ntt_launches, 1) / info["domainSize"] if ntt_launches else 1.0), ntt_ms,
                      ntt_launches, ntt_points, ntt_mads, transforms_per_launch=(ntt_points / max(ntt_launches, 1) / info["domainSize"]) if ntt_launches else None)]
        kern = [e for e in kern if e["launches"]]
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, corrected as the microarch guide says): NOT
        # measured in this run -- read from the committed summary of the same workload under profiles/ (null if absent)
        import glob
        sources, pmc_note = [], None
        if single and not args.no_pmc and not args.pmc_summary:
            # measured in this run: the prover goes (its 95 GiB of HBM are the children's), two counter children prove the same
            # workload under rocprofv3 --pmc
            try:
                prover.close()
            except Exception:                          # noqa: BLE001
                pass
            summary, pmc_note = measure_traffic(args, log_domain)
            if summary is not None:
                sources.append((summary, pmc_note))
        if args.pmc_summary:
            sources.append((args.pmc_summary, "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload made beside this run "
                                               "(tools/run_r4_prof.sh), corrected as the microarch guide says" % args.pmc_summary))
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
            sources.append((path, "profiles/%s (separate rocprofv3 --pmc passes on this workload; NOT measured in this run%s)"
                            % (os.path.basename(path), ": " + pmc_note if pmc_note else "")))
        for src, label in sources:
            try:
                pmc = (src if isinstance(src, dict) else json.load(open(src))).get(str(log_domain), {})
            except Exception:
                continue
            measured_here = isinstance(src, dict)
            for e in kern:
                k = pmc.get(e["kernel"])
                if k and e["traffic"] is None and single and (measured_here or (args.mix == "U" and not args.g1_only)):
                    e["traffic"] = k["fetch"] + k["write"]
                    e["traffic_fetch"], e["traffic_write"] = k["fetch"], k["write"]
                    e["traffic_source"] = label
        kern.sort(key=lambda e: -e["ms_per_step"])
        roofline = dict(kern[0]) if kern else {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
        roofline["kernels"] = {e["kernel"]: e for e in kern[1:]}
        roofline["note"] = ("the kernel with the largest share of a step; integer-issue-bound kernels: issue_bound gives modmul work against the "
                            "v_mad_u64_u32 peak (DESIGN.md)")
        roofline["launch_times_from"] = ("K steps with the H branch behind the witness products (one kernel on the chip at a time), right after the "
                                         "timed region: same process, same prover" if timed_overlap else "the K timed steps themselves")
        if timed_overlap:
            roofline["launch_ms_in_timed_region"] = {n: kstats_region[i][0] for i, n in
                                                     ((3, "segment_accumulate_group_kernel<%d>" % grp_members), (1, "segment_accumulate_kernel<G2Cfg>"),
                                                      (0, "segment_accumulate_kernel<G1Cfg>"), (2, "ntt_pass_kernel")) if kstats_region[i][1]}
        roofline["board"] = board
        ms_per_step = 1e3 * elapsed / args.steps
        if clean_split is not None:
            msm_ms, fft_ms = clean_split[0] * args.steps, clean_split[1] * args.steps
        res = {
            "metric": "proofs/s", "value": args.steps / elapsed, "unit": "proofs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "value_definition": ("K / wall time of K proofs (S1-S13 of src/groth16.cpp:48-203), one after the other from one host thread, on a "
                                 "created prover with the witness already resident in HBM (ug_groth16_prover_prove_resident = groth16_prover_prove "
                                 "minus .wtns parse and PCIe copy); library switches: ULTRAGROTH_OVERLAP=%d%s" % (
                                     timed_overlap, ", ULTRAGROTH_GRAPH=1" if os.environ.get("ULTRAGROTH_GRAPH", "0") not in ("", "0") else ""))
                                if single else "K / wall time (max over ranks) of K proofs sharded over the ranks, witness resident",
            # SURVEY.md section 8(d)'s "ms/proof": the reference's own call, .wtns in HOST memory (parse + PCIe copy + device + host),
            # K calls one after the other -- measured right after the region (N = 1)
            "api_value": (1e3 / prove_call_ms) if prove_call_ms else None, "api_unit": "proofs/s",
            "api_ms_per_step": prove_call_ms,
            "api_definition": "K / wall time of K groth16_prover_prove calls (src/prover.h) on the created prover, .wtns in host memory" if prove_call_ms else None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 254-bit modular integers)",
            "data": "synthetic",
            "config": {"workload": "groth16-bn254 2^%d constraints, nVars 2^%d-1, nCoefs 4N, %s + H-poly FFT, scalar mix %s "
                                   "(BASELINE.json configs[%d] shape); step = one proof (S1-S13) on a created prover, witness resident in HBM, "
                                   "one proof after the other from one host thread"
                                   % (log_domain, log_domain, "G1 MSMs A and H only" if args.g1_only else "full G1+G2 MSM", args.mix,
                                      1 if args.g1_only else 2),
                       "log_domain": log_domain, "mix": args.mix, "b_zero": args.b_zero if single else 0.0, "overlap": timed_overlap,
                       "graph": os.environ.get("ULTRAGROTH_GRAPH", "0") not in ("", "0"),
                       "fused_g1_group": os.environ.get("ULTRAGROTH_FUSED", "1") != "0",
                       "parallelism": "one GPU" if single else "%s x%d%s" % (
                           "base-range shard" if not layout.q_log else "bucket-class shard (%d point ranges)" % len({L.witness for L in layouts}),
                           world, ", H-poly chains split over ranks" if split_h else "")},
            "msm_ms_per_proof": msm_ms / args.steps, "fft_ms_per_proof": fft_ms / args.steps,
            "unoverlapped_ms_per_step": clean_ms,
            "split_region": (("device time of the MSM (S1-S4, S10) and FFT (S5-S9) parts of the K un-overlapped steps right after the timed region "
                              "(in the region the two parts run beside each other and their stream times stretch)" if timed_overlap else
                              "device time of the MSM (S1-S4, S10) and FFT (S5-S9) parts of the K timed steps themselves") if single else
                             "stream time of rank 0's MSM and FFT parts over the K timed steps; its chains run on a second stream BESIDE "
                             "its witness MSMs, so the two overlap and their sum exceeds the step"),
            "comm": comm,
            "replicated": replicated,
            # outside the timed region (N = 1): the reference's call with the .wtns in host memory
            "prove_call_ms_per_step": prove_call_ms,
            "witness_upload_ms_per_proof": upload_ms,
            "witness_upload_gbs": (32.0 * info["nVars"] / (upload_ms * 1e-3) / 1e9) if upload_ms else None,
            "pipelined_proofs_per_s": (args.steps / pipelined_s) if pipelined_s else None,
            "pipelined_host_threads": host_threads if pipelined_s else None,
            "create_s": create_s, "zkey_bytes": zkey_bytes, "zkey_ingest_gbs": zkey_bytes / create_s / 1e9,
            "time_to_first_proof_s": first_proof_s if single else None, "tables_in_use_after_s": tables_s if single else None,
            "cold_start": ("groth16_prover_create returns once the zkey is resident (create_s); time_to_first_proof_s = create + one "
                           "groth16_prover_prove at once (classic windows, beside the window-table kernels, which run on a stream of their own); "
                           "tables_in_use_after_s = when the tables were finished and adopted; the timed region runs on them") if single else None,
            "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1048576.0, 2),
            "roofline": roofline,
        }
        # the CPU baseline is a figure of the N = 1 line (run contract: "on rank 0 at N = 1 only"); the N > 1 lines of a scaling run
        # carry null and say why, and no rank waits for forty seconds of OpenMP
        if world > 1:
            res["cpu_baseline"] = None
            res["cpu_baseline_note"] = "measured at N = 1 only (run contract); see the N = 1 line of the same build"
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(dev, args, log_domain, (zkey, wtns) if zkey is not None else None)
        if args.check:
            from oracle import closed_form
            if zkey is None:                     # N > 1: only now, and only on rank 0, the whole zkey is made (for its coefficient section)
                zkey = synth.build_circuit(dev, log_domain, mix=args.mix, g1_only=args.g1_only)[0]
            exp = closed_form.groth16_expected(zkey, wtns, synth.SEEDS, synth.g1_generator_record(), synth.g2_generator_record(),
                                               int.from_bytes(bytes(range(1, 32)), "little"), int.from_bytes(bytes(range(31, 62)), "little"),
                                               g1_only=args.g1_only,
                                               b_zero_mask=synth.b_zero_mask(info["nVars"], args.b_zero) if (single and args.b_zero) else None)
            ok = (chk[0], chk[1]) == exp
            res["check"] = "bit-exact" if ok else "MISMATCH"
            rc = 0 if ok else 3
        print(json.dumps(res))
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
