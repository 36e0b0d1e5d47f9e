"""Runs the C++ host-layer test program (tests/host/test_packet_wave.cpp over
mrnet_amd/host/PacketWave.h and the C ABI) the way the reference's driver runs its tests:
execute, then look for failures in the output (tests/mrnet_tests.sh:119-130)."""
import os
import subprocess

import pytest

from wavegen import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_packet_wave_program():
    exe = os.path.join(ROOT, "tests", "host", "test_packet_wave")
    assert os.path.exists(exe), "build it with `make` (it travels with the tree, prebuilt)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "MRNTEST_FAILURE" not in r.stdout
    assert r.stdout.count("MRNTEST_SUCCESS") >= 64


def test_c_abi_program():
    """tests/host/test_c_abi.c: the C ABI used from plain C99 (gcc)."""
    exe = os.path.join(ROOT, "tests", "host", "test_c_abi")
    assert os.path.exists(exe), "build it with `make` (it travels with the tree, prebuilt)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "MRNTEST_FAILURE" not in r.stdout
    assert r.stdout.count("MRNTEST_SUCCESS") == 8


def test_filter_plugin_program():
    """integration/mrpdr_filters.C, the MRNet filter plugin, run without MRNet
    (tests/host/test_plugin.cpp supplies the Packet methods it calls): mrpdr_Sum / Min / Max / Avg over
    waves of 1 to 70,000 packets (the large ones gathered by the plugin's helper threads), five types,
    little-endian, big-endian and mixed senders, waves with one packet of another format; and mrpdr_ArrayConcat
    and mrpdr_IntEqClass over waves of numeric and string arrays / of class packets."""
    exe = os.path.join(ROOT, "tests", "host", "test_plugin")
    assert os.path.exists(exe), "build it with `make` (it travels with the tree, prebuilt)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "MRNTEST_FAILURE" not in r.stdout
    assert r.stdout.count("MRNTEST_SUCCESS") == 4 * 3 * 5 * 4 + 4 + 1 + 4 * 3 * 6 + 1 + 4 * 3


def test_rccl_combine_program():
    """tests/host/test_rccl_combine.cpp: ncclCommInitAll over the visible devices, one thread per device,
    the nodes of mrpdr_wave_filter_partial gathered (ncclAllGather) or relayed (ncclSend / ncclRecv /
    ncclBroadcast) and combined; every rank's result against the reference's loops on the host.  One
    device here: a communicator of one rank, RCCL on device memory all the same."""
    exe = os.path.join(ROOT, "tests", "host", "test_rccl_combine")
    assert os.path.exists(exe), "build it with `make` (it travels with the tree, prebuilt)"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=200, env=env)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "MRNTEST_FAILURE" not in r.stdout
    assert "RCCL " in r.stdout and r.stdout.count("MRNTEST_SUCCESS") >= 8
