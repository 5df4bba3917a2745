#!/usr/bin/env python3
"""Two host threads driving two samplers (own models / handles) on one device, many fresh-sampler + replay rounds: every
result must equal what the sampler gives alone, and no call may fail (tests/test_hip_parity.py holds a short cut of this).
    python tools/stress_threads.py [--reps 150]
"""
import argparse
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2ms_amd import synth  # noqa: E402
from t2ms_amd.sampler import Sampler  # noqa: E402
from model.denoiser.transformer import Transformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=150)
    ap.add_argument("--unserialised", action="store_true",
                    help="DIAGNOSIS: round 4's locking -- only Sampler.run takes the per-device lock, creating / staging / destroying "
                         "samplers run concurrently with the other thread's open capture (use with T2S_LIB = a library built with "
                         "-DT2S_DIAG_UNSERIALISED: tools/variant.sh t2s_sampler diag_unser -DT2S_DIAG_UNSERIALISED).  Records WHICH "
                         "call fails with WHICH HIP error when the serialisation of DESIGN 4.5 is taken away.")
    ap.add_argument("--foreign-sync", choices=["global", "relaxed"], default=None,
                    help="a third thread calls torch.cuda.synchronize() in a loop (a caller's own device-wide call, outside every "
                         "lock): in HIP's default capture mode ('global': refused while a capture is open, and the capture dies) or "
                         "after hipThreadExchangeStreamCaptureMode(relaxed) in that thread")
    ap.add_argument("--fresh-handles", type=int, default=0, metavar="N",
                    help="every N-th round builds a new Transformer and a new LA-VAE (their HIP handles are created inside the loop)")
    a = ap.parse_args()
    if a.unserialised:
        import contextlib
        from t2ms_amd import sampler as S
        real_lock = S._run_lock
        S._run_lock = lambda device: contextlib.nullcontext()

        def run_with_lock(self, text, x_T=None, noise=None, decode=True, trace=False):
            with real_lock(self.device):
                return self._run_locked(text, x_T, noise, decode, trace)
        S.Sampler.run = run_with_lock
        S._destroy_locked = lambda device_key, ptr: S.L.lib().t2s_sampler_destroy(ptr)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _, vae = bench.build_models(dev)
    models = []
    for seed in (31337, 4242):
        m = Transformer()
        m.load_state_dict(synth.make_dit_state_dict(seed, gain=0.7), strict=True)
        models.append(m.to(dev).eval())
    texts = [synth.make_text_embeddings(21 + i, 64).to(dev) for i in range(2)]
    want = []
    for i in range(2):
        s = Sampler(models[i], vae.decoder, "ddpm", 5, 9.0, 64, 48, dev, seed=100 + i, lanes=2)
        lat, ser, _ = s.run(texts[i])
        want.append((lat.clone(), ser.clone()))
    torch.cuda.synchronize(dev)
    bad, errors = [0, 0], []

    def work(i):
        try:
            torch.cuda.set_device(dev)
            model, dec = models[i], vae.decoder
            for rep in range(a.reps):
                if a.fresh_handles and rep % a.fresh_handles == a.fresh_handles - 1:
                    # a new model and a new LA-VAE: torch uploads on the default stream, t2s_dit_create, the bf16x3 weight
                    # packing and t2s_vae_create, all while the other thread may have a capture open
                    model = Transformer()
                    model.load_state_dict(synth.make_dit_state_dict((31337, 4242)[i], gain=0.7), strict=True)
                    model = model.to(dev).eval()
                    _, v2 = bench.build_models(dev)
                    dec = v2.decoder
                s = Sampler(model, dec, "ddpm", 5, 9.0, 64, 48, dev, seed=100 + i, lanes=2 if rep % 3 else 1)
                for _ in range(2):
                    lat, ser, _ = s.run(texts[i])
                    if not (torch.equal(lat, want[i][0]) and torch.equal(ser, want[i][1])):
                        bad[i] += 1
            torch.cuda.synchronize(dev)
        except Exception as e:                       # noqa: BLE001
            import traceback
            errors.append((i, repr(e)[:300], traceback.format_exc()[-1500:]))

    stop, refused = threading.Event(), [0, 0]

    def foreign():
        """A thread of the CALLER's that keeps synchronising the device: HIP fails the call while a sampler run has a capture
        open and invalidates that capture; the runs must survive it (t2s_sampler_run re-captures)."""
        torch.cuda.set_device(dev)
        if a.foreign_sync == "relaxed":
            # what such a thread can do about it: take ITSELF out of HIP's capture bookkeeping.  In the default (global) mode
            # a thread's device-wide calls are checked against every open capture of the process; in relaxed mode they are not.
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            mode = C.c_int(2)                        # hipStreamCaptureModeRelaxed
            assert hip.hipThreadExchangeStreamCaptureMode(C.byref(mode)) == 0
        while not stop.is_set():
            try:
                torch.cuda.synchronize(dev)
                refused[1] += 1
            except Exception:                        # noqa: BLE001
                refused[0] += 1
            stop.wait(0.0005)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    fth = threading.Thread(target=foreign) if a.foreign_sync else None
    if fth:
        fth.start()
    [t.start() for t in th]
    [t.join() for t in th]
    stop.set()
    if fth:
        fth.join()
        print("foreign torch.cuda.synchronize() [%s]: refused" % a.foreign_sync, refused[0], "of", sum(refused))
    print("mismatches", bad, "errors", errors)
    sys.exit(1 if (sum(bad) or errors) else 0)


if __name__ == "__main__":
    main()
