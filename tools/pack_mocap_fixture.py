"""Packs a sample of the reference's A1 mocap frames (datasets/mocap_motions_a1/*.txt: data, not code) into tests/golden/mocap_a1_frames.npz:
every 4th frame of every clip, columns 7:19 (joint angles) and 19:31 (toe positions in the base frame, as the reference's retargeting tool
computed them from its kinematic model of a1.urdf; layout: motion_loader.py:26-48).  Values unchanged (fp32 of the 5-decimal text).
Run in the build container (needs /root/reference):   python tools/pack_mocap_fixture.py"""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sorted(glob.glob("/root/reference/datasets/mocap_motions_a1/*.txt"))
assert len(files) == 13, files
joint, toe, clip = [], [], []
for i, f in enumerate(files):
    fr = np.array(json.load(open(f))["Frames"], np.float32)[::4]
    assert fr.shape[1] == 61
    joint.append(fr[:, 7:19]); toe.append(fr[:, 19:31]); clip += [i] * len(fr)
out = os.path.join(ROOT, "tests", "golden", "mocap_a1_frames.npz")
np.savez_compressed(out, joint_pos=np.concatenate(joint), toe_pos_base=np.concatenate(toe), clip=np.array(clip, np.int16),
                    clip_names=np.array([os.path.basename(f) for f in files]))
print(out, np.concatenate(joint).shape, os.path.getsize(out), "bytes")
