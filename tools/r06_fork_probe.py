"""Probe: which fork / join patterns a HIP-graph capture of this stack accepts (side launches forked from inside autograd's backward)."""
import sys, torch, faulthandler
faulthandler.enable()
dev = torch.device("cuda:0")
which = sys.argv[1]
streams = {}


def side_of(main):
    k = 0 if which == "shared" else main.cuda_stream
    if k not in streams:
        streams[k] = torch.cuda.Stream(dev)
    return streams[k]


pending = []


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return a * w

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        main = torch.cuda.current_stream(dev)
        if torch.cuda.is_current_stream_capturing() and which != "nofork":
            side = side_of(main)
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                gw = (g * a).sum().reshape(1)
            g.record_stream(side); a.record_stream(side)
            pending.append(side)
        else:
            gw = (g * a).sum().reshape(1)
        return g * w, gw


def join():
    cur = torch.cuda.current_stream(dev)
    while pending:
        cur.wait_stream(pending.pop())


ws = [torch.ones(1, device=dev, requires_grad=True) for _ in range(8)]
x = torch.ones(1 << 16, device=dev)
br = torch.cuda.Stream(dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
TWO = ("two", "nofork", "joinmain", "flow", "shared", "prefork")
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    main = torch.cuda.current_stream()
    a = x * 2
    if which == "prefork":
        side_of(br).wait_stream(main)
    for w in ws[:4]:
        a = F.apply(a, w)
    if which in TWO:
        br.wait_stream(main)
        with torch.cuda.stream(br):
            b = a.detach() * 1.5
            for w in ws[4:]:
                b = F.apply(b, w)
            lb = b.sum()
            lb.backward()
            if which in ("two", "nofork", "shared", "prefork"):
                join()
                rb = sum(w.grad for w in ws[4:]) + 0
        if which == "joinmain":
            join()                      # the nested side stream joins MAIN
            main.wait_stream(br)
            rb = sum(w.grad for w in ws[4:]) + 0
        if which == "flow":
            w1 = pending.pop()
            while pending:
                pending.pop()
            w1.wait_stream(br)
            with torch.cuda.stream(w1):
                rb = sum(w.grad for w in ws[4:]) + 0     # the branch's tail runs ON its side stream
            main.wait_stream(w1)
    la = a.sum()
    la.backward()
    join()
    r = sum(w.grad for w in ws[:4]) + 0
    if which in TWO:
        main.wait_stream(br)
print(which, "captured"); sys.stdout.flush()
g.replay(); torch.cuda.synchronize()
print(which, "ok", float(r), float(rb) if which in TWO else None)
