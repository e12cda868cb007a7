"""Parameters whose gradient is ZERO in exact arithmetic (test infrastructure, shared by tests/test_fullsize_gpu.py
and bench.py's parity leg).

Both the HIP path and the oracle hold rounding noise there (|g| ~ 1e-8 of the model's largest gradient, 100 % apart),
so a relative error is meaningless; they are bounded ABSOLUTELY against the largest parameter gradient instead.  The
classes, each with the reason the true gradient vanishes (train mode, batch-statistics BN):

  attention_spatial_s2f.value_conv.bias   reaches the output as gamma * b_v (softmax rows sum to one) straight into the
                                          batch-statistics bn_s2f, which subtracts the batch mean
                                          (wdf_attention_helper.py:41-54, custom_video_model_builder.py:143-146)
  attention_spatial_s2f.key_conv.bias     adds q.b_k to every score of a query's row: softmax is shift-invariant
  attention_channel_f2s.conv.weight       ECA's gate scales each channel in front of bn_f2s, which divides that scale
                                          out again (up to eps) (wdf_attention_helper.py:77-91)
  banch2.4.bias / banch1.1.bias           ShuffleNetV2: the BN behind the depthwise conv; its bias is a per-channel
                                          constant into a bias-free 1x1x1 conv + batch-statistics BN
                                          (shufflenetv2_helper.py:62-64, 74-75, 89-91)
  shortcut.1.bias / bn_dw.bias            GhostNet: the same situation in the shortcut (depthwise conv, BN, 1x1x1 conv,
                                          BN) and behind the stride-2 depthwise conv when no SqueezeExcite sits between
                                          it and ghost2's primary 1x1x1 conv + BN (ghostnet_helper.py:114-143)
  s5 ... ghost2.cheap_operation.1.bias,   GhostNet's last stage: a per-channel constant on the block output only ever
  s5 ... shortcut.3.bias                  meets 1x1x1 convs + batch-statistics BNs (the next block's primary conv, the
                                          identity shortcuts, the head's ConvBnAct)

A name pattern alone does not exclude a parameter: the oracle's own gradient must also be below 1e-5 of the model's
largest parameter-gradient norm (bn_dw.bias in a block WITH SqueezeExcite is a real gradient and stays in the
statistic)."""
import re

PATTERNS = [re.compile(p) for p in (
    r"attention_spatial_s2f\.(value|key)_conv\.bias$",
    r"attention_channel_f2s\.conv\.weight$",
    r"\.banch2\.4\.bias$", r"\.banch1\.1\.bias$",
    r"\.shortcut\.1\.bias$", r"\.bn_dw\.bias$",
    r"^s5\..*\.(ghost2\.cheap_operation\.1|shortcut\.3)\.bias$",
)]
# how many parameters fall into the classes above at the BASELINE shapes (bench.py --workload <key>): a pattern that
# starts to swallow real gradients — or a renamed module that drops out of the list — changes the count
EXPECTED = {"dual": 12, "slowfast": 0, "ghostnet": 38, "shufflenetv2": 46}
NOISE_REL = 1e-5     # |oracle gradient| below this fraction of the largest gradient norm ...
ABS_BOUND = 2e-5     # ... and then |hip - oracle| must stay below this fraction of it


def split(ref_grads):
    """ref_grads: {name: oracle gradient}.  Returns (noise names, largest gradient norm)."""
    gmax = max(float(g.norm()) for g in ref_grads.values())
    noise = [k for k, g in ref_grads.items()
             if any(p.search(k) for p in PATTERNS) and float(g.norm()) < NOISE_REL * gmax]
    return noise, gmax


def check_noise(got, ref_grads, noise, gmax):
    """Every analytically-zero gradient of the HIP path is rounding noise too: absolute bound."""
    for k in noise:
        d = float((got[k].double() - ref_grads[k].double()).norm())
        assert d < ABS_BOUND * gmax, (k, float(got[k].norm()), float(ref_grads[k].norm()), gmax)
