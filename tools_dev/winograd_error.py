"""Dev (CPU): numerics of Winograd F(2x2, 3x3) with bf16 MFMA operands against the direct bf16-operand convolution --
the question the next round's plan has to answer before any kernel is written.
Direct:   bf16(x) * bf16(w), fp32 accumulate                      (what the HIP kernels do today)
Winograd: V = B^T d B computed in fp32 from bf16(x), rounded to bf16; U = G g G^T in fp32 from bf16(w), rounded to bf16;
          M = sum_c U*V in fp32; Y = A^T M A in fp32.
Reference: float64 convolution of the bf16-rounded inputs."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
def bf(t): return t.float().bfloat16().double()
def winograd(x, w):                       # x [C,H,W] (H, W even), w [N,C,3,3]; pad 1
    C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(1, 4, 2).unfold(2, 4, 2)            # [C, H/2, W/2, 4, 4]
    V = bf(Bt @ tiles @ Bt.T)                             # transform in fp32/64, operand rounded to bf16
    U = bf(G @ w @ G.T)                                   # [N, C, 4, 4]
    M = torch.einsum('nckl,cyxkl->nyxkl', U, V)           # fp32-like accumulate (exact here)
    Y = At @ M @ At.T                                     # [N, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 3, 2, 4).reshape(w.shape[0], H, W)
for C, N, H in ((64, 64, 32), (256, 64, 16), (512, 64, 16)):
    x = bf(torch.relu(torch.randn(C, H, H)))              # post-ReLU activations
    w = bf(torch.randn(N, C, 3, 3) / (9 * C) ** 0.5)
    ref = F.conv2d(x[None], w, padding=1)[0]
    direct = ref                                          # bf16 operands, exact accumulate: the reference itself
    wino = winograd(x, w)
    out_bf = lambda y: y.float().bfloat16().double()      # the stored activation is bf16 either way
    e_store = float((out_bf(ref) - ref).norm() / ref.norm())
    e_wino = float((wino - ref).norm() / ref.norm())
    print("C=%4d: bf16 storage of the result %.2e | Winograd operand rounding %.2e (x%.1f)" % (C, e_store, e_wino, e_wino / e_store))
