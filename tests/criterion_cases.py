"""The cases of tests/golden/criterion.npz (written by tools/gen_golden_criterion.py from the reference's own criterion
modules), as (key, constructor arguments) lists shared by the CPU test of the oracle and the GPU test of the HIP path."""
SDR_CASES = [(c, r, bm) for c in ("SDR", "NegSDR") for r in ("mean", "sum", None) for bm in (True, False)]
PIT_CASES = [(S, c, r) for S in (2, 3)
             for c, r in (("NegSDR", "mean"), ("SDR", "mean"), ("NegSISDR", "sum"), ("SISDR", "mean"), ("NegSDR", "sum"))]
ORPIT_CASES = ["NegSISDR", "SISDR", "NegSDR"]
SINK_CASES = [(S, c, cold, it) for S in (2, 3, 4)
              for c, cold, it in (("NegSISDR", 1.0, 10), ("NegSISDR", 0.2, 10), ("SISDR", 1.0, 5), ("NegSDR", 0.5, 10))]
