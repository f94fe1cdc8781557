"""dev probe (GPU box): do two independent chains of narrow kernels, captured on two streams into ONE hipGraph, overlap
when the graph is replayed?  Prints the replay time of (a) both chains on one stream, (b) one chain per stream (fork at the
head, join at the tail), (c) as (b) with the origin stream waiting for a token at the head of the second chain."""
import ctypes, sys, time
sys.path.insert(0, ".")
import torch
from phiseg_code_amd import runtime as rt
L = rt.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
P, C = 1024, 192
def mk():
    x = torch.randn(P, C, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    f = lambda n: torch.zeros(n, device="cuda")
    return dict(x=x, y=y, g=torch.ones(C, device="cuda"), b=f(C), m=f(C), r=f(C), sc=f(C), sh=f(C))
A, B_ = mk(), mk()
def chain(t, stream, n):
    for _ in range(n):
        L.bn_small_fwd(t["x"].data_ptr(), 1, t["g"].data_ptr(), t["b"].data_ptr(), 1e-3, t["y"].data_ptr(), t["m"].data_ptr(), t["r"].data_ptr(),
                       t["sc"].data_ptr(), t["sh"].data_ptr(), None, None, 0.0, P, C, 1, stream)
def new_stream():
    s = ctypes.c_void_p(); L.stream_create(ctypes.byref(s)); return s
def new_event():
    e = ctypes.c_void_p(); L.event_create(ctypes.byref(e)); return e
s0, s1 = new_stream(), new_stream()
chain(A, s0, 2); chain(B_, s1, 2); L.stream_sync(s0); L.stream_sync(s1)
def capture(body):
    L.stream_sync(s0); L.graph_begin_capture(s0)
    body()
    ge = ctypes.c_void_p(); L.graph_end_capture(s0, ctypes.byref(ge)); return ge
def timeit(ge, reps=20):
    for _ in range(3): L.graph_launch(ge, s0)
    L.stream_sync(s0); t0 = time.perf_counter()
    for _ in range(reps): L.graph_launch(ge, s0)
    L.stream_sync(s0); return (time.perf_counter() - t0) / reps * 1e6
def serial(): chain(A, s0, N); chain(B_, s0, N)
def forked(token):
    def body():
        ef, et, ej = new_event(), new_event(), new_event()
        L.event_record(ef, s0); L.stream_wait_event(s1, ef)
        if token:
            chain(B_, s1, 1); L.event_record(et, s1); L.stream_wait_event(s0, et)
        chain(A, s0, N); chain(B_, s1, N - (1 if token else 0))
        L.event_record(ej, s1); L.stream_wait_event(s0, ej)
    return body
def late_dep():
    # second chain: first half independent, second half waits for the END of the first chain (the prior's latent levels
    # wait for the posterior's samples); ideal: N launches + N/2, serialised: 2N
    ef, em, ej = new_event(), new_event(), new_event()
    L.event_record(ef, s0); L.stream_wait_event(s1, ef)
    chain(A, s0, N); chain(B_, s1, N // 2)
    L.event_record(em, s0); L.stream_wait_event(s1, em)
    chain(B_, s1, N - N // 2)
    L.event_record(ej, s1); L.stream_wait_event(s0, ej)
def mid_dep():          # event recorded in the MIDDLE of the origin chain; both streams have work before and after
    ef, em, ej = new_event(), new_event(), new_event()
    L.event_record(ef, s0); L.stream_wait_event(s1, ef)
    chain(A, s0, N // 2); chain(B_, s1, N // 2)
    L.event_record(em, s0); L.stream_wait_event(s1, em)
    chain(A, s0, N - N // 2); chain(B_, s1, N - N // 2)
    L.event_record(ej, s1); L.stream_wait_event(s0, ej)
def serial_two_streams():   # the second stream only starts after the first chain: nothing can overlap
    ef, ej = new_event(), new_event()
    chain(A, s0, N)
    L.event_record(ef, s0); L.stream_wait_event(s1, ef)
    chain(B_, s1, N)
    L.event_record(ej, s1); L.stream_wait_event(s0, ej)
def one(): chain(A, s0, N)
_tb = torch.zeros(8, dtype=torch.int64, device="cuda")
def stamps():
    for _ in range(N): L.stamp(_tb.data_ptr(), s0)
print("chain of %d 1-thread launches:    %8.1f us  (pure per-node cost)" % (N, timeit(capture(stamps))))
print("one chain of %d launches:        %8.1f us" % (N, timeit(capture(one))))
print("two chains, one stream:          %8.1f us" % timeit(capture(serial)))
print("two chains, two streams:         %8.1f us" % timeit(capture(forked(False))))
print("two streams, 2nd waits mid-way:  %8.1f us  (ideal 1.5 chains)" % timeit(capture(late_dep)))
print("two streams, event mid-chain:    %8.1f us  (ideal 1 chain)" % timeit(capture(mid_dep)))
print("second stream after the first:   %8.1f us  (ideal 2 chains)" % timeit(capture(serial_two_streams)))
print("two chains, two streams + token: %8.1f us" % timeit(capture(forked(True))))
