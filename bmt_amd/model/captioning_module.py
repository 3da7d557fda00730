"""Drop-in for the reference's model/captioning_module.py: Transformer :16-98, BiModalTransformer :101-187."""
import torch
import torch.nn as nn

from .. import ops
from .blocks import (BridgeConnection, FeatureEmbedder, Identity, PositionalEncoder, VocabularyEmbedder, _embed)
from .decoders import BiModelDecoder, Decoder
from .encoders import BiModalEncoder, Encoder
from .generators import Generator


def _load_encoder_weights(path, strip):
    cpt = torch.load(path, map_location='cpu', weights_only=False)
    weights = {k: v for k, v in cpt['model_state_dict'].items() if 'encoder' in k}
    return cpt['config'], {k.replace(strip, ''): v for k, v in weights.items()}


class Transformer(nn.Module):
    """Uni-modal captioning model (--modality audio|video), reference :16-98."""

    def __init__(self, train_dataset, cfg):
        super(Transformer, self).__init__()
        self.register_load_state_dict_post_hook(ops.weights_changed)   # cached bf16 weight planes go stale
        self.modality = cfg.modality

        if cfg.modality == 'video':
            self.d_model = cfg.d_model_video
            self.d_feat = cfg.d_vid
            self.d_ff = cfg.d_ff_video
        elif cfg.modality == 'audio':
            self.d_feat = cfg.d_aud
            self.d_model = cfg.d_model_audio
            self.d_ff = cfg.d_ff_audio

        if cfg.use_linear_embedder:
            self.src_emb = FeatureEmbedder(self.d_feat, self.d_model)
        else:
            assert self.d_feat == self.d_model
            self.src_emb = Identity()

        self.trg_emb = VocabularyEmbedder(train_dataset.trg_voc_size, self.d_model)
        self.pos_emb = PositionalEncoder(self.d_model, cfg.dout_p)
        self.encoder = Encoder(self.d_model, cfg.dout_p, cfg.H, self.d_ff, cfg.N)
        self.decoder = Decoder(self.d_model, cfg.dout_p, cfg.H, self.d_ff, cfg.N)
        self.generator = Generator(self.d_model, train_dataset.trg_voc_size)

        print('initialization: xavier')
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.trg_emb.init_word_embeddings(train_dataset.train_vocab.vectors, cfg.unfreeze_word_emb)

        if cfg.pretrained_prop_model_path is not None:
            print(f'Pretrained prop path: \n {cfg.pretrained_prop_model_path}')
            encoder_config, encoder_weights = _load_encoder_weights(cfg.pretrained_prop_model_path, 'encoder.')
            if cfg.modality == 'video':
                self.d_model = encoder_config.d_model_video
                self.d_ff = encoder_config.d_ff_video
            elif cfg.modality == 'audio':
                self.d_model = encoder_config.d_model_audio
                self.d_ff = encoder_config.d_ff_audio
            self.encoder = Encoder(self.d_model, encoder_config.dout_p, encoder_config.H, self.d_ff, encoder_config.N)
            self.encoder.load_state_dict(encoder_weights)
            self.encoder = self.encoder.to(cfg.device)
            for param in self.encoder.parameters():
                param.requires_grad = cfg.finetune_prop_encoder

    def forward(self, src: dict, trg, masks: dict):
        """one modality's features (B, T, d_feat) + caption prefix (B, T_c) under a key-padding mask (B, 1, T) and the caption's causal mask (B, T_c, T_c) -> log-probabilities (B, T_c, vocabulary)"""
        if self.training:
            ops.rng_advance()
        if self.modality == 'audio':
            src, add = src['audio'], None
            src_mask = masks['A_mask']
        elif self.modality == 'video':
            src, add = src['rgb'], src['flow']
            src_mask = masks['V_mask']
        trg_mask = masks['C_mask']

        if isinstance(self.src_emb, Identity):
            src = self.pos_emb(src, fuse_add=add)
        else:
            src = self.pos_emb(self.src_emb(src if add is None else src + add))
        trg = self.pos_emb(self.trg_emb(trg))

        memory = self.encoder(src, src_mask)
        out = self.decoder(trg, memory, src_mask, trg_mask)
        return self.generator(out)


class BiModalTransformer(nn.Module):
    """The flagship model of the path (reference model/captioning_module.py:101-187): features of two modalities in, per-token log-probabilities out.

    ``forward(src, trg, masks)``: ``src`` is the dict of padded feature stacks the dataset hands over -- 'rgb' and 'flow' (B, T_v, d_vid), summed
    before anything else, and 'audio' (B, T_a, d_aud); ``trg`` the caption prefix (B, T_c) of token ids; ``masks`` the dict built by make_masks
    ('V_mask' / 'A_mask': key-padding masks (B, 1, T), 'C_mask': padding AND causality (B, T_c, T_c)).  Returns (B, T_c, vocabulary) log-probabilities.
    What runs underneath: the valid rows of the two streams packed (ops.RowPack), encoder and decoder on libbmt_hip.so's kernels through
    bmt_amd.ops, the first decoder layer's self-attention on a third stream beside the encoder."""

    def __init__(self, cfg, train_dataset):
        super(BiModalTransformer, self).__init__()
        self.register_load_state_dict_post_hook(ops.weights_changed)   # cached bf16 weight planes go stale

        if cfg.use_linear_embedder:
            self.emb_A = FeatureEmbedder(cfg.d_aud, cfg.d_model_audio)
            self.emb_V = FeatureEmbedder(cfg.d_vid, cfg.d_model_video)
        else:
            self.emb_A = Identity()
            self.emb_V = Identity()

        self.emb_C = VocabularyEmbedder(train_dataset.trg_voc_size, cfg.d_model_caps)

        self.pos_enc_A = PositionalEncoder(cfg.d_model_audio, cfg.dout_p)
        self.pos_enc_V = PositionalEncoder(cfg.d_model_video, cfg.dout_p)
        self.pos_enc_C = PositionalEncoder(cfg.d_model_caps, cfg.dout_p)

        self.encoder = BiModalEncoder(
            cfg.d_model_audio, cfg.d_model_video, cfg.d_model, cfg.dout_p, cfg.H,
            cfg.d_ff_audio, cfg.d_ff_video, cfg.N
        )

        self.decoder = BiModelDecoder(
            cfg.d_model_audio, cfg.d_model_video, cfg.d_model_caps, cfg.d_model, cfg.dout_p,
            cfg.H, cfg.d_ff_caps, cfg.N
        )

        self.generator = Generator(cfg.d_model_caps, train_dataset.trg_voc_size)

        print('initialization: xavier')
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        # (order matters for bit-identical initial weights: xavier over every matrix first, then the word vectors -- model/captioning_module.py:139-145)
        self.emb_C.init_word_embeddings(train_dataset.train_vocab.vectors, cfg.unfreeze_word_emb)

        if cfg.pretrained_prop_model_path is not None:
            print(f'Pretrained prop path: \n {cfg.pretrained_prop_model_path}')
            encoder_config, encoder_weights = _load_encoder_weights(cfg.pretrained_prop_model_path, 'encoder.')
            self.encoder = BiModalEncoder(
                encoder_config.d_model_audio, encoder_config.d_model_video, encoder_config.d_model,
                encoder_config.dout_p, encoder_config.H, encoder_config.d_ff_audio,
                encoder_config.d_ff_video, encoder_config.N
            )
            self.encoder.load_state_dict(encoder_weights)
            self.encoder = self.encoder.to(cfg.device)
            for param in self.encoder.parameters():
                param.requires_grad = cfg.finetune_prop_encoder

    def row_packs(self, src: dict, masks: dict):
        """(audio RowPack, video RowPack) when this forward pass can run on PACKED ROWS (ops.PACK_ROWS: the valid positions of the two
        streams compacted, padded positions never computed -- exact, see bmt_amd.ops), else None: the streams enter the encoder straight
        from the feature stacks (no embedder in between), and every attention that would meet packed rows -- the encoder's, the
        decoder's encoder-decoder attentions -- runs on the kernels that take them (d_k 128 / 256, a one-pass operand format)."""
        A = src['audio']
        if not (ops.PACK_ROWS and ops.FUSE_RESIDUAL and ops.LN_PLANES_ONLY and A.is_cuda and isinstance(self.emb_V, Identity)
                and isinstance(self.emb_A, Identity) and hasattr(self.encoder, "encoder_AV") and hasattr(self.decoder, "decoder")):
            return None
        if ops.context().kv_cache is not None or any(t.requires_grad for t in src.values()):
            return None
        atts = [a for l in self.encoder.encoder_AV.layers for a in (l.self_att_M1, l.self_att_M2, l.bi_modal_att_M1, l.bi_modal_att_M2)]
        atts += [a for l in self.decoder.decoder.layers for a in (l.enc_att_A, l.enc_att_V)]
        for a in atts:
            if a.d_k not in (128, 256) or ops.policy_of(a).attn == ops.PREC_BF16X3:
                return None
        return ops.pack_rows(masks['A_mask']), ops.pack_rows(masks['V_mask'])

    def encode(self, src: dict, masks: dict, packs=None):
        """features -> encoder memory (Av, Va): rgb + flow, (optional) linear embedders, positional tables, dropout, bi-modal
        encoder (reference :165-181).  Independent of the caption prefix: greedy decoding calls it once (bmt_amd.decode).
        packs = row_packs(...): the two streams (and the memories returned) hold packed rows."""
        A = src['audio']
        if packs is not None:
            V = self.pos_enc_V(src['rgb'], fuse_add=src['flow'], pack=packs[1])
            A = self.pos_enc_A(A, pack=packs[0])
        elif isinstance(self.emb_V, Identity):
            V = self.pos_enc_V(src['rgb'], fuse_add=src['flow'])
            A = self.pos_enc_A(A)
        else:
            V = self.pos_enc_V(self.emb_V(src['rgb'] + src['flow']))
            A = self.pos_enc_A(self.emb_A(A))
        return self.encoder((A, V), masks)

    def embed_caption(self, trg):
        C = trg
        if isinstance(self.emb_C.embedder, nn.Embedding):
            pc = self.pos_enc_C
            C = _embed(self.emb_C, C, pe=pc.table(self.emb_C.embedder.weight.device),
                       p=pc.dout_p if self.training else 0.0, site=pc._site)
        else:
            C = self.pos_enc_C(self.emb_C(C))
        return C

    def decode(self, trg, memory, masks: dict):
        """caption prefix + encoder memory -> decoder states (B, Sc, Dc)  (reference :172-173,182-184)"""
        return self.decoder((self.embed_caption(trg), memory), masks)

    def forward(self, src: dict, trg, masks: dict):
        if self.training:
            ops.rng_advance()   # every forward pass draws fresh dropout masks, as nn.Dropout does
        # the caption embedding and the first decoder layer's self-attention sublayer do not depend on the encoder: they are issued on a
        # side stream (ops.fork_side_stream) beside it -- and so is their backward, beside the encoder's
        packs = self.row_packs(src, masks)
        s3 = ops.fork_side_stream(1) if (trg.is_cuda and hasattr(self.decoder, "decoder")) else None
        if s3 is None:
            memory = self.encode(src, masks, packs)
            C = self.decode(trg, memory, masks)
            return self.generator(C)
        s1 = torch.cuda.current_stream()
        first = self.decoder.decoder.layers[0]
        for t in (trg, masks['C_mask']):
            t.record_stream(s3)
        with torch.cuda.stream(s3):
            C = first.self_attention_sublayer(self.embed_caption(trg), masks['C_mask'])
        memory = self.encode(src, masks, packs)
        s1.wait_stream(s3)
        C.record_stream(s1)
        C._bmt_self_att_done = True
        return self.generator(self.decoder((C, memory), masks))
