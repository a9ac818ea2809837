"""Checkpoint wire format (trainval_net_voc.py:488-500 writer, :307-319 / test_net_coco.py:275-279
readers): a file in the reference's layout, holding a state_dict with the REFERENCE's key names
and shapes (oracle/detector_ref.reference_shapes, enumerated from the imported reference), loads
strictly into the build's model; save -> load round-trips model, optimizer, session and epoch."""
import torch

from oracle import detector_ref as D


def _model():
    from ait_amd.faster_rcnn import resnet
    m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    return m


def test_reference_layout_checkpoint_loads_and_round_trips(tmp_path):
    from ait_amd import checkpoint, config
    shapes = D.reference_shapes()
    ref_sd = D.make_detector_state_dict(5, shapes)
    m = _model()
    own = m.state_dict()
    # the reference's file also carries BN num_batches_tracked etc.; fill what the generator
    # does not produce from the model itself so that the load below can be strict
    full = {k: ref_sd.get(k, v) for k, v in own.items()}
    assert set(shapes) <= set(own) and all(tuple(own[k].shape) == tuple(shapes[k]) for k in shapes)
    f = tmp_path / checkpoint.checkpoint_name("", "voc", "res50", 1, 3, 99).lstrip("/")
    torch.save({"session": 1, "epoch": 4, "model": {"module." + k: v for k, v in full.items()},
                "pooling_mode": "align", "class_agnostic": True}, f)
    config.cfg.POOLING_MODE = "crop"
    session, epoch, lr = checkpoint.load_checkpoint(str(f), m)
    assert (session, epoch, lr) == (1, 4, None) and config.cfg.POOLING_MODE == "align"
    k = "transformer.encoder.layer_stack.0.slf_attn.w_qs.weight"
    assert torch.equal(m.state_dict()[k], ref_sd[k])

    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.004, momentum=0.9)
    params[0].grad = torch.ones_like(params[0])
    opt.step()                                        # creates a momentum buffer
    g = tmp_path / "own.pth"
    checkpoint.save_checkpoint(str(g), torch.nn.DataParallel(m), opt, session=2, epoch=6)
    raw = torch.load(g, weights_only=False)
    assert set(raw) == {"session", "epoch", "model", "optimizer", "pooling_mode", "class_agnostic"}
    assert raw["epoch"] == 7 and not any(k.startswith("module.") for k in raw["model"])
    m2 = _model()
    opt2 = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=0.1, momentum=0.9)
    session, epoch, lr = checkpoint.load_checkpoint(str(g), m2, opt2)
    assert (session, epoch, lr) == (2, 7, 0.004)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert len(opt2.state) == 1
