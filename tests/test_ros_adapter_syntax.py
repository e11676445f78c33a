"""The ROS-1 adapter (ros1/obstacle_detection_mi355x.cpp) cannot be built here (no catkin, no roscpp), so it is at least
parsed and type-checked on every run: g++ -fsyntax-only against declarations of the few roscpp / sensor_msgs names it uses
(tests/mocks/ros1, signatures as roscpp documents them) and against the real C ABI header include/pcop.h - a call into the
ABI with the wrong arguments, a missing field of pcop_node_clouds or a typo fails here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_ros1_adapter_parses_and_type_checks_against_the_abi():
    src = os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "ros1", "obstacle_detection_mi355x.cpp")
    out = subprocess.run(["g++", "-std=c++14", "-Wall", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "mocks", "ros1"),
                          "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "error" not in out.stderr
    text = open(src).read()
    # the node's surface (od.cpp:1001-1010): every topic of the reference is advertised, with its queue size
    for topic, queue in (("voxel_grid", 1), ("statistical_outliers", 1), ("indices_cloud", 1000), ("planar_cloud", 1000), ("cloud_f", 1000),
                         ("centroids", 1), ("euc_clusters", 5)):
        assert '("%s", %d)' % (topic, queue) in text, topic


def test_ros1_adapter_hands_the_message_blob_to_the_library():
    """The node-order branch must not unpack the message on the host (round 3's did: 10^8 iterations a callback, no check of steps,
    offsets, endianness or the blob's size): it calls pcop_process_node_pointcloud2, which validates the geometry (tests/test_gpu_node_chain.py
    feeds it malformed messages), and checks the blob's size itself, where the vector is."""
    src = open(os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "ros1", "obstacle_detection_mi355x.cpp")).read()
    assert "pcop_process_node_pointcloud2(" in src
    assert "cloud_msg->data.size() < (size_t)cloud_msg->height * cloud_msg->row_step" in src
    assert "cloud_msg->is_bigendian" in src.split("pcop_process_node_pointcloud2(")[1][:600]
    assert "std::memcpy(&in[" not in src   # (the per-point loop is gone)
    host = open(os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "host", "obstacle_path.cpp")).read()
    assert "pcop_process_node_pointcloud2(" in host
