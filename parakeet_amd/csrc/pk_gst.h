// pk_gst.h -- global style tokens of TransformerTTS (transformer_tts.py:299-310, :586-588; modules/style_encoder.py):
// a reference spectrogram per utterance -> one style vector of adim floats, added to every encoder output row.
// Definitions in gst.hip.
#pragma once
#include <vector>

#include "pk_fft.h"

constexpr int PK_GST_MAX_CONV = 8;

struct pk_gst_cfg {
    int idim = 80;   // mel bins of the reference spectrogram (odim of the model)
    int tokens = 10, token_dim = 256, heads = 4;
    int conv_layers = 6, conv_chans[PK_GST_MAX_CONV] = {32, 32, 64, 64, 128, 128, 0, 0};
    int conv_kernel_size = 3, conv_stride = 2;
    int gru_layers = 1, gru_units = 128;
};

struct pk_gst {
    pk_gst_cfg cfg;
    // weights (offsets into the owner's arena)
    size_t conv_w[PK_GST_MAX_CONV] = {0}, conv_b[PK_GST_MAX_CONV] = {0};   // BatchNorm2D folded: [Cout][Cin][k][k], [Cout]
    int conv_f[PK_GST_MAX_CONV + 1] = {0};                                 // frequency bins before layer i
    std::vector<size_t> gru_wih, gru_whh, gru_bih, gru_bhh;                // per layer: [in][3H], [H][3H], [3H], [3H]
    size_t stl_k = 0, stl_v = 0, stl_wq = 0, stl_bq = 0, stl_wo = 0, stl_bo = 0;
    // per call
    pk_dbuf d_a, d_b, d_tab, d_seq0, d_seq1, d_ref;
};

// validates the hyper-parameters (PK_EUNSUPPORTED / PK_EINVAL with the reason in pk_last_error)
int pk_gst_check(const pk_gst_cfg& c);
// packs the parameters under `prefix` ("gst"): ref_enc.convs.{3i}.weight, ref_enc.convs.{3i+1}.{weight,bias,_mean,_variance},
// ref_enc.gru.{weight_ih_l{l}, ... | {l}.cell.weight_ih, ...}, stl.gst_embs, stl.mha.linear_{q,k,v,out}.{weight,bias}
int pk_gst_finalize(pk_fft_arena& ar, const pk_param_map& P, const std::string& prefix, pk_gst& g);
// StyleEncoder.forward for B reference spectrograms: speech HOST packed (sum(lens), idim), lens HOST (B);
// d_style DEVICE [B][token_dim].  Launches on h->ctx->stream.
int pk_gst_run(pk_fft_core* h, pk_gst& g, const float* speech, const int* lens, int B, float* d_style);
void pk_gst_release(pk_gst& g);
