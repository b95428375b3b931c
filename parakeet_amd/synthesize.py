"""End-to-end batched synthesis: FastSpeech2 (or SpeedySpeech) -> Parallel WaveGAN on one GPU.

The device-side equivalent of the loop body of
examples/fastspeech2/ljspeech/synthesize_e2e.py:88-102
(``mel = fastspeech2_inference(phone_ids); wav = pwg_inference(mel)``), for a
ragged batch and with the mel never leaving HBM.  The only host<->device
traffic inside a call is the token ids in and B frame counts out (the output
length is data dependent).  With a ``SpeedySpeechInference`` as the acoustic model (``tones=`` per utterance) it is
the loop body of examples/speedyspeech/baker/synthesize_e2e.py:113-131; that recipe's vocoder has hop 300
(upsample_scales [4, 5, 3, 5]), which the Parallel WaveGAN kernels handle like any hop from 32 to 1024.
"""
import numpy as np
import torch

from .runtime import wrap


class Synthesizer:
    def __init__(self, fastspeech2_inference, pwg_inference, acoustic_lanes=()):
        self.am_inference, self.voc_inference = fastspeech2_inference, pwg_inference
        self.am = fastspeech2_inference.acoustic_model
        self.voc = pwg_inference.pwg_generator
        self.hop = self.voc.upsample_factor
        # issue-ahead lanes (see issue_acoustic): lane 0 is the acoustic model itself; further lanes are wrappers around OTHER
        # engine handles holding the same weights, each with its own side stream
        self._lanes = [fastspeech2_inference] + list(acoustic_lanes)
        self._lane_streams = {}

    def add_acoustic_lane(self, inference):
        """Another acoustic-model wrapper (its own engine handle, the same weights) for ``issue_acoustic(..., lane=k)``;
        returns k."""
        self._lanes.append(inference)
        return len(self._lanes) - 1

    def synthesize_packed(self, texts, alpha=1.0, noise=None, generator=None, tones=None, spk_ids=None):
        """Returns (packed wav device tensor, frames per utterance).  ``spk_ids``: one speaker id per utterance for a
        multi-speaker FastSpeech2 (examples/fastspeech2/aishell3/synthesize_e2e.py:90-98)."""
        self.am_inference.bind()
        self.voc_inference.bind()
        if type(self.am).__name__ == "SpeedySpeech":
            assert alpha == 1.0, "SpeedySpeech has no speed control (speedyspeech.py:178-218)"
            frames = self.am.encode_batch(texts, tones)
        else:
            assert tones is None, "tone ids go to FastSpeech2 through encode_batch(tone_ids=...)"
            frames = self.am.encode_batch(texts, alpha) if spk_ids is None else self.am.encode_batch(texts, alpha, spk_ids)
        if int(frames.sum()) == 0:
            return torch.empty(0, device=self.am._ctx.device), frames
        mel = self.am.decode_packed(denormalize=True)       # FastSpeech2Inference: log-mel domain
        keep = frames > 0  # the vocoder needs >= 1 frame per utterance
        wav = self.voc.infer_packed(mel, frames[keep], noise=noise, generator=generator, normalize=True)
        return wav, frames

    # ---- issue-ahead pipeline ------------------------------------------------------------------------------------
    # The frame-count sync of the acoustic model drains the stream, and its decoder's ~200 launches are then issued
    # while the GPU waits (0.2 - 4.4 ms per batch depending on the host).  For a sequence of
    # batches the acoustic model of batch k + 1 can be issued on a side stream while the vocoder of batch k runs:
    #     pending = s.issue_acoustic(batch0)
    #     for nxt in batches[1:] + [None]:
    #         wav, frames = s.vocode_issued(pending, noise); pending = s.issue_acoustic(nxt) if nxt else None
    # Results are bit-identical to synthesize_packed (same kernels, same order per handle).
    #
    # Two lanes (round 6, profiles/r06_pipeline_lanes.txt): with ONE handle the acoustic model of batch k + 1 can only be issued
    # once the vocoder of batch k is queued -- its encoder then sits behind 40 ms of persistent layer kernels, the frame-count
    # sync returns after them, and the decoder's launches are issued while the GPU waits (0.4 ms of holes per step in the
    # kernel trace).  With a second handle (same weights) on its own side stream the order can be
    #     nxt = s.issue_acoustic(batch[k + 1], lane=(k + 1) & 1); wav, frames = s.vocode_issued(pending, noise); pending = nxt
    # : batch k + 1's encoder and decoder run next to batch k's decoder and the vocoder's small kernels, and every host
    # latency is covered by queued work of the other lane.

    def issue_acoustic(self, texts, alpha=1.0, tones=None, spk_ids=None, lane=0):
        """Acoustic model of one batch on a side stream; returns (mel, frames, event) for vocode_issued.  Arguments as
        ``synthesize_packed`` (``spk_ids`` for a multi-speaker FastSpeech2); ``lane`` selects the engine handle / side
        stream (0 = the acoustic model itself, others: ``add_acoustic_lane``)."""
        inf = self._lanes[lane]
        am = inf.acoustic_model
        if lane not in self._lane_streams:
            self._lane_streams[lane] = torch.cuda.Stream(device=am._ctx.device)
        stream = self._lane_streams[lane]
        with torch.cuda.stream(stream):
            inf.bind()
            if type(am).__name__ == "SpeedySpeech":
                assert alpha == 1.0, "SpeedySpeech has no speed control (speedyspeech.py:178-218)"
                frames = am.encode_batch(texts, tones)
            else:
                assert tones is None, "tone ids go to FastSpeech2 through encode_batch(tone_ids=...)"
                frames = am.encode_batch(texts, alpha) if spk_ids is None else am.encode_batch(texts, alpha, spk_ids)
            mel = am.decode_packed(denormalize=True) if int(frames.sum()) else None
            ev = torch.cuda.Event()
            ev.record(stream)
        return mel, frames, ev

    def vocode_issued(self, issued, noise=None, generator=None):
        """Vocoder (on the caller's stream) for a batch whose acoustic model was issued with issue_acoustic."""
        mel, frames, ev = issued
        if mel is None:
            return torch.empty(0, device=self.am._ctx.device), frames
        self.voc_inference.bind()
        cur = torch.cuda.current_stream(self.am._ctx.device)
        cur.wait_event(ev)
        mel.record_stream(cur)
        keep = frames > 0
        wav = self.voc.infer_packed(mel, frames[keep], noise=noise, generator=generator, normalize=True)
        return wav, frames

    def synthesize_batch(self, texts, alpha=1.0, noises=None, generator=None, tones=None, spk_ids=None):
        noise = None
        if noises is not None:
            noise = torch.cat([torch.as_tensor(np.asarray(n)).reshape(-1) for n in noises])
        wav, frames = self.synthesize_packed(texts, alpha, noise, generator, tones, spk_ids)
        outs, o = [], 0
        for f in frames:
            n = int(f) * self.hop
            outs.append(wrap(wav[o:o + n].reshape(n, 1)))
            o += n
        return outs

    def __call__(self, text, alpha=1.0, noise=None):
        return self.synthesize_batch([text], alpha, None if noise is None else [noise])[0]


class ARSynthesizer:
    """An autoregressive acoustic model -> WaveFlow on one GPU, for a ragged batch: the loop body of
    examples/transformer_tts/synthesize.py:78-88 (``mel = transformer_tts_inference(text); wav = vocoder.infer(mel)``)
    and of the Tacotron2 + WaveFlow notebook (examples/tacotron2/synthesize.ipynb), with the mel staying in HBM.
    ``acoustic`` is a ``TransformerTTSInference`` or a ``Tacotron2``; ``vocoder`` a ``ConditionalWaveFlow``."""

    def __init__(self, acoustic, vocoder):
        self.acoustic, self.vocoder = acoustic, vocoder

    def mels(self, texts, seeds=None, **kw):
        """List of (L_b, n_mels) log-mel device tensors."""
        if type(self.acoustic).__name__ == "Tacotron2":
            outs = self.acoustic.infer_batch(texts, seeds=seeds, **kw)
            return [o["mel_outputs_postnet"] for o in outs]
        m = self.acoustic.bind()
        outs = m.inference_batch(texts, seeds=seeds, return_att=False, denormalize=True, **kw)
        return [o[0] for o in outs]

    def synthesize_batch(self, texts, seeds=None, zs=None, generator=None, **kw):
        """Lists of token ids -> list of (T_b,) waveforms (device tensors).  ``seeds``: dropout-stream seed per
        utterance; ``zs``: WaveFlow latents per utterance (else drawn by the engine / ``generator``)."""
        mels = self.mels(texts, seeds=seeds, **kw)
        # waveflow's input is (C_mel, T) per utterance (synthesize.py:81-83)
        return self.vocoder.infer_batch([m.as_subclass(torch.Tensor).transpose(0, 1) for m in mels], zs, generator)

    def __call__(self, text, seed=0, z=None, **kw):
        return self.synthesize_batch([text], [seed], None if z is None else [z], **kw)[0]
