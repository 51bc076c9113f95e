#!/usr/bin/env python3
"""Stage A of VERDICT r4 item 1: does PARTITIONING the chip (streams made by hipExtStreamCreateWithCUMask) let the HBM-bound
sparse kernels run beside the power-limited GEMMs?  Config-3 train steps (B = 65536, 26 x 1M ids, E = 64, [512, 256, 128],
bounded catch-up — the bench's configuration), no kernel change: launches are re-routed to masked streams from outside.

  part 0  where a masked stream's workgroups land (tools/probe/libcumask_where.so): the mask bit -> (XCD, CU) mapping
  part 1  every big kernel of the step ALONE on a stream of c CUs (fork + join around each launch): time over c
  part 2  (i)  the weight-gradient batch on k CUs  ||  mi_sparse_apply_fused on 256 - k   (independent today)
  part 3  (ii) forward + data-gradient GEMMs on k CUs  ||  an extra catch-up of the same size on shadow tables on 256 - k
          (what a catch-up made a step AHEAD would cost the step): none / serialised on the whole chip / beside
  part 4  (i) + (ii) together: projected step = that step - the head-of-step catch-up it would replace
  part 5  a streaming copy and a tiny launch on c CUs (no model)
  part 6  the WHOLE step with one masked queue as its stream (continuously busy: how every kernel scales over CU count)
  part 7  what a hand-over between two queues costs: pooled / default / masked in every pairing (no model)
  part 8  (iii) an extra weight-gradient batch on k CUs beside the head of the step (catch-up + gather on 256 - k)
--dedicated 1 (default): the step's main and side streams are dedicated full-mask queues too (see part 7).
Results: profiles/r05_cumask_probe.md.  Masked streams are destroyed after each configuration: more than ~6 alive in one
process make every kernel of the process erratic.
Usage: python tools/cumask_probe.py [--parts 0,...,8] [--ks 128,160,176,192,208,224] [--alone 256,192,...] [--steps 30]
"""
import argparse, ctypes as C, glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
import torch
from mi355x_rec.engine import DeepFM, OptimizerSpec

ap = argparse.ArgumentParser()
ap.add_argument("--parts", default="0,1,2,3,4")
ap.add_argument("--ks", default="128,160,176,192,208,224")
ap.add_argument("--alone", default="256,224,192,176,160,128,96,80,64,48,32")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--catchup", default="bounded")
ap.add_argument("--dedicated", type=int, default=1, help="1: the step's main and side streams are dedicated (full-mask) queues too: "
                "a hand-over between a pooled HIP stream and a masked one costs ~60 us, between two dedicated ones ~10 (part 7)")
args = ap.parse_args()
PARTS = {int(x) for x in args.parts.split(",")}
KS = [int(x) for x in args.ks.split(",")]

hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("device: %s, %d CUs" % (torch.cuda.get_device_name(0), NCU), flush=True)


def masked_stream(bits):
    """a torch stream whose kernels may use the CUs whose mask bits are in `bits`"""
    words = [0] * ((NCU + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * len(words))(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    t = torch.cuda.ExternalStream(s.value)
    t._raw = s.value
    return t


def drop_streams():
    """destroy every masked stream made so far (each owns a hardware queue: keep few alive)"""
    torch.cuda.synchronize()
    for key, t in list(_streams.items()):
        if hasattr(t, "_raw"):
            hip.hipStreamDestroy(C.c_void_p(t._raw))
        del _streams[key]


_streams = {}
def lo(k):
    """the first k mask bits (bit i -> XCD i % 8 if the driver interleaves: part 0 checks)"""
    if ("lo", k) not in _streams:
        _streams[("lo", k)] = masked_stream(range(k)) if k < NCU else torch.cuda.Stream()
    return _streams[("lo", k)]
def hi(k):
    """the mask bits from k on"""
    if ("hi", k) not in _streams:
        _streams[("hi", k)] = masked_stream(range(k, NCU))
    return _streams[("hi", k)]


# ---------------------------------------------------------------- sensors (best effort)
def _sensor_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "power1_average", "power1_input"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out
SENS = _sensor_files()
class Poll:
    def __enter__(self):
        self.v = {k: [] for k in SENS}; self.stop = False
        def run():
            while not self.stop:
                for k, p in SENS.items():
                    try: self.v[k].append(float(open(p).read()))
                    except Exception: pass
                time.sleep(0.02)
        self.t = threading.Thread(target=run, daemon=True); self.t.start(); return self
    def __exit__(self, *a):
        self.stop = True; self.t.join()
    def text(self):
        o = []
        if self.v.get("freq1_input"): o.append("sclk %.0f MHz" % (sum(self.v["freq1_input"]) / len(self.v["freq1_input"]) / 1e6))
        for k in ("power1_average", "power1_input"):
            if self.v.get(k): o.append("%.0f W" % (sum(self.v[k]) / len(self.v[k]) / 1e6)); break
        return ", ".join(o)


# ---------------------------------------------------------------- part 0
if 0 in PARTS:
    so = os.path.join(ROOT, "tools", "probe", "libcumask_where.so")
    if os.path.exists(so):
        w = C.CDLL(so)
        w.cumask_where.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        out = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda")
        def where(stream, label):
            out.zero_(); torch.cuda.synchronize()
            rc = w.cumask_where(out.data_ptr(), 4096, 200000, stream.cuda_stream)
            torch.cuda.synchronize()
            v = out.cpu().numpy().reshape(-1, 2)
            xcc = v[:, 0] & 0xF
            cu = {}
            for x, h in zip(xcc.tolist(), v[:, 1].tolist()):
                # HW_ID (gfx9): [11:8] CU, [12] SH, [15:13] SE
                cu.setdefault(x, set()).add(((h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15))
            print("  %-22s rc=%d  CUs per XCD: %s  total %d" % (label, rc, " ".join("%d:%d" % (x, len(cu[x])) for x in sorted(cu)),
                                                            sum(len(c) for c in cu.values())), flush=True)
        print("part 0: where workgroups land", flush=True)
        where(torch.cuda.Stream(), "plain stream")
        for k in (8, 64, 128, 192):
            where(lo(k), "bits [0,%d)" % k)
            where(hi(k), "bits [%d,%d)" % (k, NCU))
        where(masked_stream(range(0, 32)), "bits [0,32)")
        where(masked_stream(range(0, NCU, 8)), "bits 0,8,16,...")
        drop_streams()
    else:
        print("part 0 skipped: %s not built" % so)

# ---------------------------------------------------------------- part 5: a streaming copy and a tiny launch on c CUs
if 5 in PARTS:
    print("part 5: torch copy of 2 GiB (read + write = 4 GiB) and a 4-KiB copy on a stream of c CUs", flush=True)
    src = torch.empty(1 << 29, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    tiny_s, tiny_d = torch.zeros(1024, device="cuda"), torch.zeros(1024, device="cuda")
    for c in [int(x) for x in args.alone.split(",")]:
        s_ = lo(c)
        with torch.cuda.stream(s_):
            for _ in range(3): dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): dst.copy_(src)
            e1.record(); e1.synchronize()
            big = e0.elapsed_time(e1) / 10
            e0.record()
            for _ in range(200): tiny_d.copy_(tiny_s)
            e1.record(); e1.synchronize()
            small = e0.elapsed_time(e1) / 200
        print("  c = %3d  copy %.3f ms = %.2f TB/s (%.1f GB/s per CU)   tiny launch %.1f us back to back" %
              (c, big, 2 * src.numel() * 4 / big / 1e9, 2 * src.numel() * 4 / big / 1e6 / c, small * 1e3), flush=True)
        drop_streams()
    del src, dst
    if not (PARTS - {5, 7}):
        sys.exit(0)

# ---------------------------------------------------------------- part 7: what a hand-over between two queues costs
if 7 in PARTS:
    print("part 7: ping-pong between two streams, one ~100-us kernel each per round (event wait in both directions)", flush=True)
    x = torch.ones(1 << 26, device="cuda"); y = torch.ones(1 << 26, device="cuda")
    def pingpong(A, Bs, label, rounds=200):
        def run(n):
            for _ in range(n):
                with torch.cuda.stream(A):
                    A.wait_stream(Bs); x.mul_(1.0001)
                with torch.cuda.stream(Bs):
                    Bs.wait_stream(A); y.mul_(1.0001)
        run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(rounds); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / rounds * 1e6
        # the same kernels back to back on ONE stream
        with torch.cuda.stream(A):
            for _ in range(20): x.mul_(1.0001); y.mul_(1.0001)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(rounds): x.mul_(1.0001); y.mul_(1.0001)
            torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / rounds * 1e6
        print("  %-46s %.0f us per round, the two kernels alone on A %.0f us -> %.0f us per hand-over" % (label, dt, one, (dt - one) / 2), flush=True)
    full = masked_stream(range(NCU)); m192 = masked_stream(range(192)); m64 = masked_stream(range(192, NCU))
    pingpong(torch.cuda.Stream(), torch.cuda.Stream(), "plain <-> plain")
    pingpong(torch.cuda.current_stream(), torch.cuda.Stream(), "default <-> plain")
    pingpong(torch.cuda.Stream(), m192, "plain <-> masked(192)")
    pingpong(torch.cuda.current_stream(), m192, "default <-> masked(192)")
    pingpong(full, m192, "masked(all 256) <-> masked(192)")
    pingpong(m192, m64, "masked(192) <-> masked(64)")
    pingpong(full, torch.cuda.Stream(), "masked(all 256) <-> plain")
    for t in (full, m192, m64):
        hip.hipStreamDestroy(C.c_void_p(t._raw))
    if not (PARTS - {5, 7}):
        sys.exit(0)

# ---------------------------------------------------------------- model (overlap_probe.py's)
F, V, E, H, B = 26, 1_000_000, 64, [512, 256, 128], 65536
m = DeepFM([V] * F, embedding_size=E, hidden_units=H, dropout=0.1, optimizer=OptimizerSpec("Adam", 0.001), catchup=args.catchup)
g = torch.Generator(device="cuda"); g.manual_seed(1)
m.init_variables(g, lin_scale=1e-3)
NB = 48
batches = [(torch.randint(0, V, (B, F), device="cuda", dtype=torch.int32, generator=g),
            (torch.rand(B, device="cuda", generator=g) < 0.25).to(torch.uint8)) for _ in range(NB)]
_i = [0]
def step():
    i = _i[0]; _i[0] += 1
    a, b = batches[i % NB]
    return m.train_step(a, b, next_ids=batches[(i + 1) % NB][0])
for _ in range(40):
    step()
torch.cuda.synchronize()

K = m.k
GEMM_FWD_DGRAD = ("mi_dense_fwd_planes", "mi_hidden_logits_head_fused", "mi_dense_bwd_data_planes")
WGRAD = "mi_dense_bwd_weight_planes_batch"
APPLY = "mi_sparse_apply_fused"
BIG = GEMM_FWD_DGRAD + (WGRAD, APPLY, "mi_sparse_catchup", "mi_embed_fm_planes_fwd")
for n in BIG + ("mi_dense_apply",):
    getattr(K, n)                                     # bind
ORIG = {n: K.__dict__[n] for n in BIG + ("mi_dense_apply",)}
LAST = {"rows": None}
def _note_rows(fn):
    """remember which rows this step's head-of-step catch-up walks (the shadow catch-up walks the same list)"""
    def call(*a):
        if a[0] is not None:
            LAST["rows"] = (a[7], a[8])
        fn(*a)
    return call
ORIG["mi_sparse_catchup"] = _note_rows(ORIG["mi_sparse_catchup"])
def restore():
    for n, f in ORIG.items():
        K.__dict__[n] = f
restore()

def timed(nsteps, label, want=()):
    for _ in range(5): step()
    K.timers = {} if want else None
    K.timer_only = set(want) if want else None
    torch.cuda.synchronize()
    with Poll() as p:
        t0 = time.perf_counter()
        for _ in range(nsteps): loss, _ = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nsteps * 1e3
    per = {}
    if want:
        for key, evs in K.timers.items():
            per[key] = sum(s.elapsed_time(e) for s, e in evs) / nsteps * 1e3          # us per step
    K.timers = None; K.timer_only = None
    print("  %-44s %.3f ms/step  loss %.5f  %s  %s" % (label, dt, float(loss), p.text(),
                                                     " ".join("%s=%.0f" % (k.replace("mi_", ""), v) for k, v in sorted(per.items()))), flush=True)
    return dt, per

timed(args.steps, "baseline before anything is re-routed", want=BIG)
timed(args.steps, "baseline again, no timers")
if args.dedicated:
    torch.cuda.synchronize()
    FULL = masked_stream(range(NCU))
    for name in ("side_stream", "presort_stream", "wsplit_stream"):
        m._ws[name] = masked_stream(range(NCU))
    m._presorted = None
    torch.cuda.set_stream(FULL)
    timed(args.steps, "baseline, every stream a dedicated full-mask queue", want=BIG)
    timed(args.steps, "the same, no timers")
# ---------------------------------------------------------------- part 1: each big kernel alone on c CUs
if 1 in PARTS:
    print("part 1: each big kernel alone on a stream of c CUs (us per step; fork + join around every routed launch)", flush=True)
    def routed(fn, s):
        def call(*a):
            main = torch.cuda.current_stream()
            s.wait_stream(main)
            with torch.cuda.stream(s):
                fn(*a)
            main.wait_stream(s)
        return call
    for c in [int(x) for x in args.alone.split(",")]:
        s = lo(c)
        for n in BIG:
            K.__dict__[n] = routed(ORIG[n], s)
        timed(args.steps, "c = %d" % c, want=BIG)
        restore()
        drop_streams()

# ---------------------------------------------------------------- part 6: the WHOLE step on one masked stream (no re-routing)
if 6 in PARTS:
    print("part 6: the whole step with a masked stream as torch's current stream (the engine's own side streams stay plain)", flush=True)
    for c in [int(x) for x in args.alone.split(",")]:
        with torch.cuda.stream(lo(c)):
            torch.cuda.synchronize()
            timed(args.steps, "c = %d" % c, want=BIG)
            torch.cuda.synchronize()
        drop_streams()

# ---------------------------------------------------------------- parts 2-4
t2, m2, v2 = None, None, None
def shadow():
    global t2, m2, v2, last2
    if t2 is None:
        t2, m2, v2 = m.table.clone().contiguous(), torch.rand(m.R, E, device="cuda") * 1e-6, torch.rand(m.R, E, device="cuda") * 1e-10 + 1e-12
        last2 = torch.ones(m.R, dtype=torch.int32, device="cuda")

def overlapped(k, do_i, do_ii, serial_extra=False):
    """re-route: (ii) forward + data-gradient GEMMs -> G (k CUs) with the shadow catch-up on Hs (256 - k) beside them;
    (i) weight-gradient batch -> G, sparse apply -> Hs, dense apply after G."""
    G, Hs = lo(k), hi(k)
    st = {"phase": None, "ev": None}
    flags = 1 | (2 if args.catchup == "bounded" else 0)
    sp = m.sched.spec
    def extra():
        n = B * F
        uniq, nu = LAST["rows"]
        last2.fill_(m.step - 14)
        ORIG["mi_sparse_catchup"](t2, m2, v2, None, None, None, last2, uniq, nu, n, E, m.step, m.sched.table,
                                  sp.beta1, sp.beta2, sp.epsilon, flags, 1, 0)
    def gemm(fn, first):
        def call(*a):
            main = torch.cuda.current_stream()
            if first and st["phase"] != m.step:
                st["phase"] = m.step
                if serial_extra:
                    extra()
                else:
                    Hs.wait_stream(main)
                    with torch.cuda.stream(Hs):
                        extra()
            G.wait_stream(main)
            with torch.cuda.stream(G):
                fn(*a)
        return call
    def wgrad(*a):
        main = torch.cuda.current_stream()
        if do_ii:
            main.wait_stream(G)                      # d_concat and the dY planes exist
            if not serial_extra:
                main.wait_stream(Hs)
        if do_i:
            st["ev"] = torch.cuda.Event(); st["ev"].record(main)
            G.wait_stream(main)
            with torch.cuda.stream(G):
                ORIG[WGRAD](*a)
        else:
            ORIG[WGRAD](*a)
    def dense_apply(*a):
        if do_i:
            torch.cuda.current_stream().wait_stream(G)
        ORIG["mi_dense_apply"](*a)
    def apply(*a):
        if do_i:
            main = torch.cuda.current_stream()
            Hs.wait_event(st["ev"])
            with torch.cuda.stream(Hs):
                ORIG[APPLY](*a)
            main.wait_stream(Hs)
        else:
            ORIG[APPLY](*a)
    if do_ii:
        shadow()
        for j, n in enumerate(GEMM_FWD_DGRAD):
            K.__dict__[n] = gemm(ORIG[n], n == "mi_dense_fwd_planes")
    K.__dict__[WGRAD] = wgrad
    K.__dict__["mi_dense_apply"] = dense_apply
    K.__dict__[APPLY] = apply

def serial_extra_only():
    """the shadow catch-up serialised on the whole chip before the forward GEMMs (what it costs when nothing hides it)"""
    shadow()
    st = {"phase": None}
    flags = 1 | (2 if args.catchup == "bounded" else 0)
    sp = m.sched.spec
    def first(*a):
        if st["phase"] != m.step:
            st["phase"] = m.step
            n = B * F
            uniq, nu = LAST["rows"]
            last2.fill_(m.step - 14)
            ORIG["mi_sparse_catchup"](t2, m2, v2, None, None, None, last2, uniq, nu, n, E, m.step, m.sched.table,
                                      sp.beta1, sp.beta2, sp.epsilon, flags, 1, 0)
        ORIG["mi_dense_fwd_planes"](*a)
    K.__dict__["mi_dense_fwd_planes"] = first

# ---------------------------------------------------------------- part 8: an extra weight-gradient batch beside the HEAD of the step
def wgrad_beside_head(k, mode):
    """what deferring a step's weight gradients to the head of the NEXT step would cost: an EXTRA mi_dense_bwd_weight_planes_batch
    (the previous call's jobs, results to scratch) on G = k CUs while the catch-up and the gather run on H = 256 - k.
    mode: 'serial' = the extra batch on the main stream before the catch-up; 'beside' = as described; 'plain' = beside, on a
    plain (unmasked) pair of streams."""
    from mi355x_rec import _lib
    G = lo(k) if mode == "beside" else torch.cuda.Stream()
    Hs = hi(k) if mode == "beside" else torch.cuda.Stream()
    st = {"job": None, "phase": None}
    scratch = torch.empty(m.P, dtype=torch.float32, device="cuda")
    def wgrad(arr, n, Bn, ws, wsn):
        # a copy of the job list with dW / db pointing into scratch, and a workspace of its own
        a2 = (_lib.WgradJob * n)()
        for q in range(n):
            C.memmove(C.byref(a2[q]), C.byref(arr[q]), C.sizeof(_lib.WgradJob))
            a2[q].dW = scratch.data_ptr() + (arr[q].dW - m.d_grad.data_ptr())
            a2[q].db = scratch.data_ptr() + (arr[q].db - m.d_grad.data_ptr())
        if st["job"] is None or st["job"][4].numel() < wsn:
            ws2 = torch.empty(wsn, dtype=torch.uint8, device="cuda")
        else:
            ws2 = st["job"][4]
        st["job"] = (a2, n, Bn, None, ws2, wsn)
        ORIG[WGRAD](arr, n, Bn, ws, wsn)
    def extra():
        a2, n, Bn, _, ws2, wsn = st["job"]
        ORIG[WGRAD](a2, n, Bn, ws2, wsn)
    def catchup(*a):
        main = torch.cuda.current_stream()
        rows = a[0] is not None
        if rows and st["job"] is not None and mode != "none":
            st["phase"] = m.step
            if mode == "serial":
                extra()
                ORIG["mi_sparse_catchup"](*a)
                return
            G.wait_stream(main); Hs.wait_stream(main)
            with torch.cuda.stream(G):
                extra()
            with torch.cuda.stream(Hs):
                ORIG["mi_sparse_catchup"](*a)
            return
        ORIG["mi_sparse_catchup"](*a)
    def gather(*a):
        main = torch.cuda.current_stream()
        if st["phase"] == m.step and mode in ("beside", "plain"):
            Hs.wait_stream(main)
            with torch.cuda.stream(Hs):
                ORIG["mi_embed_fm_planes_fwd"](*a)
            main.wait_stream(Hs); main.wait_stream(G)
            return
        ORIG["mi_embed_fm_planes_fwd"](*a)
    K.__dict__[WGRAD] = wgrad
    K.__dict__["mi_sparse_catchup"] = catchup
    K.__dict__["mi_embed_fm_planes_fwd"] = gather

if 8 in PARTS:
    print("part 8: an extra weight-gradient batch on k CUs beside the catch-up + gather on 256 - k (weight gradients deferred to the next step's head)", flush=True)
    b8, _ = timed(args.steps, "baseline")
    wgrad_beside_head(0, "serial")
    ser, _ = timed(args.steps, "extra batch serialised on the whole chip")
    restore()
    print("      -> costs %.3f ms when nothing hides it" % (ser - b8), flush=True)
    wgrad_beside_head(0, "plain")
    dt, _ = timed(args.steps, "beside, plain streams")
    restore()
    print("      -> exposed %.3f ms; a step with its weight gradients deferred: %.3f ms (baseline %.3f)" % (dt - b8, dt - (ser - b8), b8), flush=True)
    for k in KS:
        wgrad_beside_head(k, "beside")
        dt, _ = timed(args.steps, "k = %d" % k)
        restore()
        drop_streams()
        print("      -> exposed %.3f ms; a step with its weight gradients deferred: %.3f ms (baseline %.3f)" % (dt - b8, dt - (ser - b8), b8), flush=True)

base = None
if PARTS & {2, 3, 4}:
    print("baseline (plain streams, the product's step)", flush=True)
    base, per = timed(args.steps, "baseline", want=("mi_sparse_catchup", APPLY, WGRAD))
    base2, _ = timed(args.steps, "baseline, no timers")
    base = min(base, base2)
    head_catchup = None
    # the head-of-step catch-up of the rows (the larger of the two calls under the key): its own measurement
    K.timers = {}; K.timer_only = {"mi_sparse_catchup"}
    for _ in range(10): step()
    torch.cuda.synchronize()
    ev = K.timers["mi_sparse_catchup"]; K.timers = None; K.timer_only = None
    durs = sorted(s.elapsed_time(e) for s, e in ev)
    head_catchup = sum(durs[len(durs) // 2:]) / (len(durs) - len(durs) // 2)         # the row calls are the longer half
    print("  head-of-step row catch-up: %.3f ms" % head_catchup, flush=True)

if 2 in PARTS:
    print("part 2 (i): weight-gradient batch on k CUs || sparse apply on 256 - k", flush=True)
    for k in KS:
        overlapped(k, True, False)
        dt, _ = timed(args.steps, "k = %d" % k)
        restore()
        drop_streams()
        print("      -> %+.3f ms vs baseline %.3f" % (dt - base, base), flush=True)

if 3 in PARTS:
    print("part 3 (ii): forward + data-gradient GEMMs on k CUs || an extra (shadow) catch-up on 256 - k", flush=True)
    serial_extra_only()
    ser, _ = timed(args.steps, "extra catch-up serialised, whole chip")
    restore()
    print("      -> costs %.3f ms when nothing hides it" % (ser - base), flush=True)
    for k in KS:
        overlapped(k, False, True)
        dt, _ = timed(args.steps, "k = %d" % k)
        restore()
        drop_streams()
        print("      -> exposed %.3f ms of %.3f (%.0f %% hidden); a step with the catch-up made AHEAD: %.3f ms" %
              (dt - base, ser - base, 100 * (1 - (dt - base) / max(ser - base, 1e-9)), dt - head_catchup), flush=True)

if 4 in PARTS:
    print("part 4: (i) + (ii)", flush=True)
    for k in KS:
        overlapped(k, True, True)
        dt, _ = timed(args.steps, "k = %d" % k)
        restore()
        drop_streams()
        print("      -> %.3f ms; with the head-of-step catch-up gone: %.3f ms (baseline %.3f)" % (dt, dt - head_catchup, base), flush=True)
