"""-m gpu: the supervised_train driver (flags / loop / log format / stats files of the reference) end to end."""
import os
import re

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [["--model", "graphsage_mean"], ["--model", "gcn", "--sampler", "padded", "--max_degree", "16"],
                                   ["--model", "graphsage_maxpool", "--sigmoid"],
                                   ["--model", "graphsage_mean", "--identity_dim", "16"],
                                   ["--model", "graphsage_mean", "--dropout", "0.2"]])
def test_supervised_train_driver(dev, tmp_path, capsys, extra):
    from graphsage_amd import engine as eng
    from graphsage_amd import supervised_train as st
    eng.reset_engine()
    f1 = st.main(["--synthetic", "small", "--epochs", "2", "--batch_size", "128", "--samples_1", "5", "--samples_2", "3",
                  "--dim_1", "32", "--dim_2", "32", "--validate_iter", "10", "--print_every", "5",
                  "--base_log_dir", str(tmp_path)] + extra)
    out = capsys.readouterr().out
    assert "Epoch: 0001" in out and "Optimization Finished!" in out and "Full validation stats:" in out
    assert re.search(r"Iter: \d{4} train_loss= \d+\.\d{5} train_f1_mic= \d\.\d{5} .* val_f1_mic= \d\.\d{5} .* time= \d+\.\d{5}", out)
    stats = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path)) for f in fs]
    assert any(p.endswith("val_stats.txt") for p in stats) and any(p.endswith("test_stats.txt") for p in stats)
    txt = open([p for p in stats if p.endswith("val_stats.txt")][0]).read()
    assert re.match(r"loss=\d+\.\d{5} f1_micro=\d\.\d{5} f1_macro=\d\.\d{5} time=\d+\.\d{5}", txt)
    assert f1 > (0.3 if "--sigmoid" in extra else 0.6), f1


def test_supervised_driver_device_path_equals_host_feed_path(dev, tmp_path, capsys):
    """The driver's default path (epoch order + labels resident in HBM, steps between printed lines replayed without a
    host round trip, short last batch included) prints the reference's log format and trains like the per-step
    feed_dict path."""
    from graphsage_amd import engine as eng
    from graphsage_amd import supervised_train as st
    f1s, outs = [], []
    for path in ("device", "host"):
        eng.reset_engine()
        f1s.append(st.main(["--synthetic", "small", "--epochs", "3", "--batch_size", "100", "--samples_1", "5", "--samples_2", "3",
                            "--dim_1", "32", "--dim_2", "32", "--validate_iter", "7", "--print_every", "3", "--feed_path", path,
                            "--base_log_dir", str(tmp_path / path)]))
        outs.append(capsys.readouterr().out)
    lines = [[l for l in o.splitlines() if l.startswith("Iter:")] for o in outs]
    assert len(lines[0]) == len(lines[1]) > 5                      # same cadence of printed iterations
    assert [l.split()[1] for l in lines[0]] == [l.split()[1] for l in lines[1]]
    assert abs(f1s[0] - f1s[1]) < 0.08 and min(f1s) > 0.6, f1s
