"""Host-side logic of the product that needs no GPU: buffers, configuration, launch heuristics."""
import numpy as np
import torch


def test_positional_tables_equal_the_reference(golden):
    """the sinusoid table is built in float64 and cast (Models.py:33-45); the
    product's buffers must equal the reference's bit for bit (golden g1)."""
    from ait_amd.system import PositionalEncoding, Transformer
    g = golden("g1_pos_table")
    assert np.array_equal(PositionalEncoding(512, n_position=64).pos_table[0].numpy(), g["pos_table_64_512"])
    assert np.array_equal(PositionalEncoding(64, n_position=200).pos_table[0].numpy(), g["pos_table_200_64"])
    t = Transformer(d_k=64, d_v=64, d_model=512, d_word_vec=512, d_inner=2048, n_position=64,
                    n_layers=1, n_head=8, dropout=0.1)
    for coder in (t.encoder, t.decoder):
        assert np.array_equal(coder.position_enc.pos_table[0].numpy(), g["pos_table_64_512"])
    assert "encoder.position_enc.pos_table" in dict(t.named_buffers())
    assert "encoder.position_enc.pos_table" not in dict(t.named_parameters())


def test_split_k_heuristic_is_a_multiple_of_the_xcd_count():
    from ait_amd.system import _split_k
    for (m, n, k) in [(512, 2048, 76800), (2048, 512, 76800), (1536, 512, 76800), (512, 64, 76800),
                      (1024, 512, 58800), (512, 512, 256), (2048, 512, 19200)]:
        s = _split_k(m, n, k)
        assert s % 8 == 0 and 8 <= s <= 128
        assert k // s >= 256 or s == 8


def test_cfg_from_list_follows_the_reference_convention():
    from ait_amd import config
    saved = config.cfg.TRAIN.BATCH_SIZE
    try:
        config.cfg_from_list(['TRAIN.BATCH_SIZE', 300])
        assert config.cfg.TRAIN.BATCH_SIZE == 300 and config.cfg['TRAIN'].BATCH_SIZE == 300
        try:
            config.cfg_from_list(['TRAIN.NO_SUCH_KEY', 1])
            raise AssertionError("unknown key accepted")
        except KeyError:
            pass
    finally:
        config.cfg.TRAIN.BATCH_SIZE = saved


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the hot path raises on a non-GPU tensor instead of computing somewhere else."""
    import pytest
    from ait_amd import _lib
    with pytest.raises(_lib.AitHipError):
        _lib.dev_ptr(torch.zeros(4))
