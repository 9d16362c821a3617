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
