"""Generates REFERENCE golden vectors by importing the reference's own model.py -- if it can run.

The reference (daigo0927/pwcnet) is TensorFlow-1.x graph code (`tensorflow-gpu==1.8.0`,
reference requirements.txt:63).  TensorFlow is not installed in the build image and cannot be
installed (no network), so today this script stops at its first check and writes nothing: the
oracle stays PARITY UNPINNED (oracle/pwc_oracle.c header).  The day an
environment has TF 1.x (or TF 2.x with `tensorflow.compat.v1`), running

    python tests/golden/make_reference_golden.py [--reference /root/reference]

builds the reference graph (`PWCDCNet` from the reference's model.py, imported from where it
lies -- nothing is copied), assigns the SAME seeded weights the tests use
(tests/util.model_weights: names are the reference's variable names), runs the reference forward
on the seeded inputs of the committed oracle goldens and stores inputs + outputs as
`tests/golden/ref_e2e_64x128_dc{0,1}.npz`.  tests/test_oracle.py::test_oracle_vs_reference_golden
picks those files up when they exist and then pins the oracle to the reference itself.

Only data (inputs, outputs) is written; no reference source or bytecode is stored.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def import_tf_v1():
    try:
        import tensorflow as tf
    except ImportError:
        return None
    if hasattr(tf, "compat") and hasattr(tf.compat, "v1") and not hasattr(tf, "variable_scope"):
        tf = tf.compat.v1
        tf.disable_v2_behavior()
    return tf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    tf = import_tf_v1()
    if tf is None:
        print("tensorflow is not importable here: no reference vectors written (oracle stays parity-unpinned)")
        return 1
    if not os.path.exists(os.path.join(args.reference, "model.py")):
        print(f"{args.reference}/model.py not found")
        return 1
    sys.path.insert(0, args.reference)
    sys.modules.setdefault("tensorflow", sys.modules[tf.__name__.split(".")[0]])
    import model as ref_model          # the reference's model.py, imported in place

    from tests import util
    im0, im1 = util.smooth_images(1, 64, 128)          # the inputs of e2e_64x128_dc*.npz
    for use_dc in (False, True):
        w = util.model_weights(use_dc)
        tf.reset_default_graph()
        a = tf.placeholder(tf.float32, (1, 64, 128, 3))
        b = tf.placeholder(tf.float32, (1, 64, 128, 3))
        net = ref_model.PWCDCNet(use_dc=use_dc)
        flows_final, flows_pyramid = net(a, b)
        with tf.Session(config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
            sess.run(tf.global_variables_initializer())
            seen = set()
            for v in net.vars:
                key = v.name.split(":")[0]
                if key in w:
                    sess.run(v.assign(w[key]))
                    seen.add(key)
            missing = sorted(set(w) - seen)
            assert not missing, f"reference graph lacks variables {missing[:4]}"
            final, pyr = sess.run([flows_final, flows_pyramid], {a: im0, b: im1})
        out = {"images_0": im0, "images_1": im1, "flows_final": final}
        for l, p in enumerate(pyr):
            out[f"flows_{l}"] = p
        path = os.path.join(HERE, f"ref_e2e_64x128_dc{int(use_dc)}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, "max |flows_final| =", float(np.abs(final).max()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
