"""The one-kernel first layer with its dropout decisions drawn in line against drawn ahead (dcr_dropout_words_dev), at the C
entry point: times of both, of the drawing kernel, and a check that the drawn words are what the kernel reads (all-ones
words under a valid stamp must change the output).  N, F, H, C from the environment."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]
import torch
from dcr import _lib
from models import gcn
dev = torch.device('cuda', 0)
n, F, H, C = int(os.environ.get('N', 1000000)), int(os.environ.get('F', 256)), int(os.environ.get('H', 128)), int(os.environ.get('C', 16))
g = torch.Generator(device=dev).manual_seed(0)
F16 = (F + 15) // 16 * 16
ax = torch.zeros(n, F16, device=dev)
ax[:, :F] = torch.randn(n, F, device=dev, generator=g)
w1 = torch.randn(H, F, device=dev, generator=g) * F ** -0.5
b1 = torch.randn(H, device=dev, generator=g) * 0.1
w2 = torch.randn(C, H, device=dev, generator=g) * 0.1
L = _lib.lib()
words = ctypes.c_int64()
_lib.check(L.dcr_relu_dropout_bits_words(n * H, ctypes.byref(words)))
bits = torch.zeros(words.value, dtype=torch.int64, device=dev)
_lib.check(L.dcr_dropout_words_count(n, H, ctypes.byref(words)))
dwords = torch.zeros(words.value, dtype=torch.int64, device=dev)
pre = torch.empty(n, H, device=dev)
both = torch.empty(n, 2 * C, device=dev)
cur = torch.cuda.current_stream(dev).cuda_stream
st = ctypes.c_void_p(cur)
ws = gcn._first_layer_workspace(dev, cur, n, F, H)
ws_ptr, ws_n = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())
p, seed = 0.5, 1234


def fwd(dw, off=7):
    _lib.check(L.dcr_first_layer_fwd_ws_f32_dev(ax.data_ptr(), F16, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), pre.data_ptr(), both.data_ptr(),
                                                both.data_ptr() + 4 * C, 2 * C, bits.data_ptr(), None if dw is None else dw.data_ptr(), n, F, H, C, p,
                                                seed, off, None, ws_ptr, ws_n, st))


def draw(off=7):
    _lib.check(L.dcr_dropout_words_dev(dwords.data_ptr(), n, H, p, seed, off, None, st))


def tm(f, name):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    reps = int(os.environ.get('REPS', 20))
    for _ in range(reps):
        f()
    ev[1].record()
    torch.cuda.synchronize()
    print(f'{name}: {ev[0].elapsed_time(ev[1]) / reps * 1e3:.1f} us', flush=True)


fwd(None)
ref, ref_bits = both.clone(), bits.clone()
draw()
fwd(dwords)
print('drawn == in line:', torch.equal(both, ref) and torch.equal(bits, ref_bits))
print('next offset slot:', int(dwords[-4].item()), 'stamp:', dwords[-8:-4].tolist())
dwords[:-8] = -1
fwd(dwords)
print('all-ones words change the output (the words are read):', not torch.equal(both, ref))
fwd(dwords, off=8)
fwd(None, off=8)
r8 = both.clone()
fwd(dwords, off=8)
print('another offset draws in line:', torch.equal(both, r8))
draw()
tm(lambda: fwd(None), 'forward, decisions in line')
tm(lambda: fwd(dwords), 'forward, decisions drawn ahead')
tm(draw, 'the drawing kernel')
