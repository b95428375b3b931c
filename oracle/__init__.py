"""CPU oracle for the Parakeet synthesis hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / numpy, fp32 with an fp64
switch) of the arithmetic that PaddlePaddle/Parakeet executes on the
synthesis hot path named in BASELINE.json:

    FastSpeech2.inference  ->  PWGGenerator.inference      (headline path)
    ConditionalWaveFlow.infer                              (alternative vocoder)
    STFT -> mel -> log10                                   (metric / feature path)
    SpeedySpeech.inference, TransformerTTS.inference, Tacotron2.infer   (the other acoustic models)

Every function cites the reference file:line it follows (paths relative to
the reference repository root).

Who may import this package: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` -- always as the *checker* or as the
timed CPU baseline, never as the thing that is shipped or measured as the
product.  ``parakeet_amd`` never imports it; the product path fails loudly
when the HIP library is missing.

PARITY PINNING STATUS
---------------------
The arithmetic of the reference lives in PaddlePaddle (third-party, >=2.1.2,
not vendored, not installable in this environment), and the reference's own
tests for this path hold no golden vectors (tests/unit/test_pwg.py and
test_stft.py only print; test_expansion.py asserts a shape).  The oracle is
therefore pinned in two ways, both weaker than a real Paddle run:

  1. against every machine-checkable fact the reference repository holds for
     this path (docstring worked examples, test_expansion's input/shape,
     parameter counts, shape algebra) -- tests/test_oracle_facts.py;
  2. against the reference's *own Python source* executed in this container
     over a torch-backed stand-in for the ``paddle`` API
     (oracle/paddle_shim) -- tests/golden/ holds the resulting vectors and
     tools/make_golden.py is the generating script.  This pins the op order,
     masks, transposes and layout logic of Parakeet's code; it does NOT pin
     Paddle's own kernel semantics (weight layouts, rounding of ties, eps
     defaults), which are encoded from Paddle's documentation in
     oracle/nn_ref.py and listed in DESIGN.md as "paddle-semantics,
     unverified".

The autoregressive models keep their decoder prenet's dropout on at inference; their oracles and the reference-source
runs that produced their golden vectors share the engine's counter-based dropout stream (philox_ref.dropout_keep,
injected into the stand-in's F.dropout by tools/make_golden_ar.py), so masks are identical by construction.
paddle.nn.LSTM / LSTMCell semantics (gate order i, f, g, o; aliased parameter names) are [paddle-semantics].

Until a real Paddle build has been run against these vectors the status is:
"parity pinned to the reference's Python source over a Paddle stand-in;
Paddle kernel semantics unpinned".

How to change that status (one command, on a machine with paddlepaddle >= 2.1.2 and a checkout of the reference):

    PARAKEET_REAL_PADDLE=1 PARAKEET_REFERENCE=/path/to/Parakeet python tools/verify_with_paddle.py \
        --fs2-ckpt fastspeech2_nosil_ljspeech_ckpt_0.5 --pwg-ckpt pwg_ljspeech_ckpt_0.5

tools/ref_import.py then leaves the stand-in OFF sys.path and the same generators run over Paddle itself into
tests/golden_paddle/; the script prints a per-tensor diff against the stand-in vectors, and every golden test, the
checkpoint-reader tests and tests/test_released_ckpt_*.py then compare this oracle (and, with -m gpu, the engine) with
what Paddle computed -- including on the released LJSpeech checkpoints.  tests/test_verify_paddle_cpu.py runs the very
same script over the stand-in.
"""
