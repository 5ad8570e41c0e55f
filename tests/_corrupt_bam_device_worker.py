"""Worker of tests/test_inflate.py::test_device_reader_survives_corrupted_files (own process: a GPU fault or a crash is an exit
code). The device reader must either equal the host reader, or give up in favour of it, or raise. Usage: SEED N"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from _corrupt_bam_worker import corrupt, inflate_all, source_bam, write_raw  # noqa: E402
from volcanosv_amd import bam  # noqa: E402
from volcanosv_amd.abi import VsvError  # noqa: E402
from volcanosv_amd.engine import Engine  # noqa: E402

if __name__ == "__main__":
    import tempfile
    rng = np.random.default_rng(int(sys.argv[1]))
    d = tempfile.mkdtemp()
    raw = inflate_all(source_bam(d))
    same = fell_back = raised = host_only_error = 0
    warnings.simplefilter("ignore")
    with Engine(0) as eng:
        for it in range(int(sys.argv[2])):
            p = os.path.join(d, "c.bam")
            write_raw(p, corrupt(raw, rng), block=int(rng.choice([3000, 60000])))
            host = None
            try:
                with bam.BamFile(p) as bf:
                    host = bf.fetch_soa(None)
            except (VsvError, KeyError, ValueError, OSError):
                pass
            try:
                with bam.BamFile(p) as bf:
                    view = bf.fetch_device(eng, None, sa=bool(it % 2))      # every other file also collects the SA:Z texts
                    if isinstance(view, bam.DeviceRecordView):
                        dev = view.to_host()
                        if host is None:
                            host_only_error += 1          # e.g. a malformed tag the device reader never looks at
                        else:
                            for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar"):
                                assert np.array_equal(getattr(host, name), getattr(dev, name)), (it, name)
                            if it % 2:
                                assert list(view.sa_tags) == list(host.sa_tags), it
                            same += 1
                    else:
                        fell_back += 1
            except (VsvError, KeyError, ValueError, OSError):
                raised += 1
    print("same %d fell_back %d raised %d host_only_error %d" % (same, fell_back, raised, host_only_error))
