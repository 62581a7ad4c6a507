"""(Bits 2, 4 and 32 acted on the fragment-ring version of the product loop measured in
profiles/r03_generic_ablation.jsonl; the current loop has no ring and ignores them.)
Ablation timing of the generic segment-plan kernels (diagnostic build -DNJ_GEN_ABL,
tools/ubench/build_stamps.sh): the same training step with parts of the ODE forward switched off
through NJODE_GEN_DBG bits (results are wrong by construction; only the kernel times are read).
  1 no activation / dropout in the epilogue   2 no MFMA (one vector op instead)
  4 no ring refills (fragments loaded once per layer)   8 no record stores   16 no ODE input / update
  32 no fragment loads at all   64 no barriers in the Euler-step loop"""
import contextlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from njode_amd import _lib, models  # noqa: E402


def w(n):
    return ((n, 'tanh'), (n, 'tanh'))


def main():
    dev = torch.device('cuda', 0)
    width = int(os.environ.get('ABL_WIDTH', '100'))
    for B in (100,):
        b, meta = bench.make_batch(B, seed=1)
        cfg = dict(bench.model_cfg(0.1), ode_nn=w(width), enc_nn=w(width), readout_nn=w(width))
        args = (b['times'], b['time_ptr'], b['X'].to(dev), b['obs_idx'].to(dev, torch.int32),
                meta['dt'], meta['maturity'], b['start_X'].to(dev), b['n_obs_ot'].to(dev, torch.int32))
        with contextlib.redirect_stdout(sys.stderr):
            torch.manual_seed(0)
            m = models.NJODE(**cfg).to(dev).train()
        for bits in (0, 4, 4 | 32, 64, 31, 31 | 32, 31 | 32 | 64):
            os.environ['NJODE_GEN_DBG'] = str(bits)
            for _ in range(2):
                m.loss_and_grad(*args)
            torch.cuda.synchronize()
            _lib.profile_enable(1)
            for _ in range(3):
                m.loss_and_grad(*args)
            _lib.profile_enable(False)
            k = {n: round(v[1] / max(v[0], 1), 4) for n, v in _lib.profile_read().items()}
            print(json.dumps({'width': width, 'B': B, 'dbg_bits': bits,
                              'k_gseg_ode_fwd_ms': k.get('k_gseg_ode_fwd'), 'k_gseg_mid_ms': k.get('k_gseg_mid'),
                              'k_gseg_enc_ms': k.get('k_gseg_enc')}), flush=True)


if __name__ == '__main__':
    main()
