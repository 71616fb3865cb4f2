"""The drop-in engine's ce_args (scenario file `ce_args` -> argv, reference: src/crts.cpp:43-81; getopt idiom of
cognitive_engines/CE_Template/CE_Template.cpp:17-25) on the GPU, through tests/harness/engine_harness — the ECR's rx / CE workers
played packet by packet: FFT size, estimator (ref / energy / welch / scan), frames per decision, thresholds, batch size, and the
trainer's weights arriving as a file.  Every decision and feature is checked against the oracle on the same bytes.
The option parsing and the control flow are covered without a GPU by tests/harness/engine_unit.cpp."""
import numpy as np
import pytest

import crnsense as cs
import oracle_py as orc
import signals
from test_gpu_parity import _run_harness

pytestmark = pytest.mark.gpu
TX = {0: 0.0, 1: 835e6, 2: 833e6, 3: 835e6}   # CE_Predictive_Node.cpp:247,252,257 (ALL BUSY: no call)


def _epoch_lines(out):
    full = [ln.split() for ln in out if ln.startswith("epoch ") and " feat " in ln]
    allp = [ln.split() for ln in out if ln.startswith("epoch ")]
    return full, allp


def _first_occupied(occ):
    d = np.zeros(occ.shape[0], np.int32)
    for e in range(occ.shape[0]):
        hits = np.nonzero(occ[e, 1:4])[0]
        d[e] = hits[0] + 1 if hits.size else 0
    return d


@pytest.mark.parametrize("n_fft,L,sync", [(1024, 364, False), (1024, 364, True), (4096, 364, False), (2048, 2048, False), (512, 363, False)])
def test_engine_energy_mode_at_every_size(built, tmp_path, n_fft, L, sync):
    """-n <fft_len> -m energy: the scaled channel plan, sum |X|^2 per band, thresholds relative to the noise-floor band; the
    engine's decision is the first occupied of CH1..CH3 (the reference's cascade order) and drives set_tx_freq like .cpp:245-261."""
    cfg = cs.cfg_energy_scaled(n_fft, 4.0)
    n_epochs = 16
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=100 + n_fft, L=L)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0", "-n", str(n_fft), "-m", "energy"] + (["-a", "0"] if sync else []), tmp_path, iq)
    full, _ = _epoch_lines(out)
    assert len(full) == n_epochs
    want = orc.run(cfg, iq, n_epochs, L=L)
    wd = _first_occupied(want["occupancy"])
    if 2 * L >= n_fft:   # (a 364-sample packet zero-padded to 4096 points smears a carrier over its neighbours: the detector, CPU or
        assert np.array_equal(wd, picks)   # GPU, then reports what the padded spectrum holds, not what was driven)
    for e, w in enumerate(full):
        assert int(w[3]) == wd[e] and float(w[5]) == TX[int(wd[e])]
        assert np.allclose([float(x) for x in w[7:11]], want["features"][e], rtol=1e-5)


def _welch_stream(cfg, n_epochs, L, seed):
    """Per epoch one contiguous run of P packets of L samples (the engine's view of a sensing period): the first (K - 1) hop + N
    samples carry the epoch's traffic, the rest of the last packet is noise."""
    N, K, hop = cfg.fft_len, cfg.frames_per_epoch, cfg.hop
    span = (K - 1) * hop + N
    P = -(-span // L)
    rng = np.random.default_rng(seed)
    picks = rng.integers(0, 4, n_epochs)
    chunks = []
    for e in range(n_epochs):
        x, _ = signals.make_epochs(cfg, 1, seed=seed * 1000 + e, picks=[int(picks[e])])
        assert x.size == span * 2
        tail = rng.normal(0, np.sqrt(0.5e-6), (P * L - span) * 2).astype(np.float32)
        chunks += [x, tail]
    return np.concatenate(chunks), picks, P


@pytest.mark.parametrize("L,sync", [(364, False), (364, True), (1024, False)])
def test_engine_welch_mode_takes_the_packets_as_one_stream(built, tmp_path, L, sync):
    """-m welch: Hann, 50 % overlap, K = 8 frames cut from the contiguous run of a sensing period's packets (the ring lays them end
    to end; the launch uses whole frames with the run's length as epoch stride) — against the oracle with the same stride."""
    n_fft, K = 1024, 8
    cfg = cs.cfg_welch_scaled(n_fft, K, 4.0)
    n_epochs = 12
    iq, picks, P = _welch_stream(cfg, n_epochs, L, seed=7)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0", "-n", str(n_fft), "-m", "welch", "-k", str(K)] + (["-a", "0"] if sync else []),
                       tmp_path, iq)
    full, _ = _epoch_lines(out)
    assert len(full) == n_epochs
    want = orc.run(cfg, iq, n_epochs, L=n_fft, epoch_stride=P * L)
    wd = _first_occupied(want["occupancy"])
    assert np.array_equal(wd, picks)
    for e, w in enumerate(full):
        assert int(w[3]) == wd[e] and float(w[5]) == TX[int(wd[e])]
        assert np.allclose([float(x) for x in w[7:11]], want["features"][e], rtol=1e-5)


def test_engine_scan_mode_calibrates_its_thresholds_at_start_up(built, tmp_path):
    """-m scan: 64 equal bands on the Welch estimate; thresholds = lambda x the noise floor measured over the first -c epochs
    (crn_noise_floor_host: the median band energy) — those epochs are not acted on; afterwards a driven
    channel is found through the bands that lie on it."""
    n_fft, K, L, calib = 1024, 8, 512, 6
    plan = cs.cfg_welch_scaled(n_fft, K, 4.0)          # traffic on the reference's channels
    n_epochs = calib + 12
    iq, picks, P = _welch_stream(plan, n_epochs, L, seed=11)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "1", "-n", str(n_fft), "-m", "scan", "-c", str(calib), "-t", "4"], tmp_path, iq)
    full, _ = _epoch_lines(out)
    assert len(full) == n_epochs - calib
    nf_line = [ln for ln in out if "noise floor" in ln and "threshold" in ln][0].split()
    nf = float(nf_line[nf_line.index("floor") + 1])
    # the same estimate from the oracle's features of the calibration epochs: lower median over epochs of each epoch's lower median
    scan = cs.cfg_welch(n_fft, K, 64)
    feats = orc.run(scan, iq, n_epochs, L=n_fft, epoch_stride=P * L)["features"]
    med = np.sort(feats[:calib], axis=1)[:, (64 - 1) // 2]
    want_nf = np.sort(med)[(calib - 1) // 2]
    assert abs(nf / want_nf - 1) < 1e-4
    for e, w in enumerate(full):
        assert int(w[3]) == picks[calib + e] and float(w[5]) == TX[int(picks[calib + e])]


def test_engine_batches_epochs_per_launch(built, tmp_path):
    """-b 4: four epochs per launch of the enqueue-only path; every decision still arrives, in order, the last incomplete batch
    at flush()."""
    cfg = cs.cfg_reference()
    L, n_epochs = 364, 10   # 2 full batches + 2 epochs left for the flush
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=515, L=L)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0", "-b", "4", "-s", "1"], tmp_path, iq)
    _, allp = _epoch_lines(out)
    assert [int(w[1]) for w in allp] == list(range(n_epochs))
    assert [int(w[3]) for w in allp] == list(picks)
    stats = [ln for ln in out if ln.startswith("CE_Predictive_Node_GPU: epochs")][0].split()
    assert int(stats[stats.index("launches") + 1]) == 3 + 1    # three batches + the constructor's warm-up launch


@pytest.mark.parametrize("n_fft", [1024, 4096])
def test_trained_weights_reach_the_engine_through_a_file(built, tmp_path, n_fft):
    """SURVEY.md §8(f)-3's loop closed at the drop-in boundary: generate -> sense (reference-mode features at this FFT size) ->
    crn_ann_train_device -> crn_cfg_save_ann -> the engine with `-n <N> -w <file>` (no recompile) decides the driven channel
    of fresh traffic packet by packet, with the fitted network's outputs equal to the oracle's forward pass of the same weights."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = cs.cfg_reference_scaled(n_fft)   # (DECIDE_ANN with the shipped weights: only the features of this launch are used)
    L, rms = 364, 0.002                    # the radio's packets; a carrier 20 dB below the level the shipped weights were fitted for
    spe = cs.samples_per_epoch(cfg, L)
    n_train = 2048
    s = cs.Sensor(cfg)
    iq = torch.zeros(n_train * spe * 2, dtype=torch.float32, device=dev)
    truth = torch.zeros(n_train, dtype=torch.int32, device=dev)
    feat = torch.zeros(n_train, 4, dtype=torch.float32, device=dev)
    sc = cs.SynthCfg()
    sc.seed, sc.noise_power, sc.signal_rms, sc.tones_per_band = 31337, 1e-6, rms, 8
    sc.pu_model, sc.signal_kind, sc.n_streams = cs.PU_UNIFORM, cs.SIG_TONES, 1
    s.synth_fill_device_ex(iq.data_ptr(), n_train, spe, sc, truth_ptr=truth.data_ptr())
    s.run_device(iq.data_ptr(), n_train, L, {"features": feat.data_ptr(), "ann_out": 0, "decision": 0, "occupancy": 0, "spectrum": 0})
    tc = cs.TrainCfg()
    tc.seed, tc.iterations, tc.restarts, tc.eta, tc.alpha, tc.normalise = 1, 400, 4, 2.0, 0.9, 1
    wih, who, loss = s.ann_train_device(tc, feat.data_ptr(), truth.data_ptr(), n_train)
    s.close()
    assert loss < 1e-3
    fitted = cs.set_ann_weights(cs.cfg_reference_scaled(n_fft), wih, who)
    wfile = tmp_path / "weights.txt"
    cs.save_ann(fitted, str(wfile))
    back = cs.load_ann(cs.cfg_reference_scaled(n_fft), str(wfile))
    assert all(back.ann_w_ih[i][j] == fitted.ann_w_ih[i][j] for i in range(5) for j in range(6))      # exact round trip
    assert all(back.ann_w_ho[j][k] == fitted.ann_w_ho[j][k] for j in range(6) for k in range(4))

    n_epochs = 24
    test_iq, picks = signals.make_epochs(fitted, n_epochs, seed=99, L=L, signal_rms=rms)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0", "-n", str(n_fft), "-w", str(wfile)], tmp_path, test_iq)
    full, _ = _epoch_lines(out)
    assert len(full) == n_epochs
    want = orc.run(fitted, test_iq, n_epochs, L=L)
    assert np.array_equal(want["decision"], picks)
    for e, w in enumerate(full):
        assert int(w[3]) == picks[e] and float(w[5]) == TX[int(picks[e])]
        assert np.allclose([float(x) for x in w[7:11]], want["features"][e], rtol=1e-5)
        assert np.abs(np.array([float(x) for x in w[12:15]]) - want["ann_out"][e]).max() < 1e-6
    # without -w the engine runs the reference's literals at this size: other network outputs on the same features (whether they
    # still decide this traffic depends on the gain: they were fitted at 512 points and one receiver gain) — the file is what changed
    out2 = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0", "-n", str(n_fft)], tmp_path, test_iq)
    full2, _ = _epoch_lines(out2)
    assert len(full2) == n_epochs
    shipped = orc.run(cs.cfg_reference_scaled(n_fft), test_iq, n_epochs, L=L)
    for e, w in enumerate(full2):
        assert np.abs(np.array([float(x) for x in w[12:15]]) - shipped["ann_out"][e]).max() < 1e-6
    assert np.abs(shipped["ann_out"] - want["ann_out"]).max() > 1e-2


@pytest.mark.parametrize("args,n_fft", [([], 512), (["-n", "1024", "-m", "energy"], 1024)], ids=["reference engine", "1024-pt energy"])
def test_cfg0_thousand_epochs_through_the_engine_surface(built, tmp_path, args, n_fft):
    """BASELINE.json configs[0] (N = 512 reference-exact and N = 1024, 3 bands + NF, K = 10, one stream,
    1000 epochs) driven through the engine / ECR test double — packet by packet behind CognitiveEngine::execute() — and compared
    epoch by epoch with the CPU restatement on the same bytes (the oracle-only form of this run is tests/test_oracle.py)."""
    # the reference's packets: 364 samples (zero-padded to 512 by .cpp:149); the 1024-point detector gets whole frames — a 364-sample
    # packet padded to 1024 points smears a carrier over its neighbours, and a relative-threshold detector then reports them too
    L, n_epochs = (364 if n_fft == 512 else n_fft), 1000
    cfg = cs.cfg_reference() if n_fft == 512 else cs.cfg_energy_scaled(n_fft, 4.0)
    iq, picks = signals.make_epochs(cfg, n_epochs, seed=2026, L=L)
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0"] + args, tmp_path, iq, timeout=300)
    full, _ = _epoch_lines(out)
    assert len(full) == n_epochs
    want = orc.run(cfg, iq, n_epochs, L=L, n_threads=8)
    wd = want["decision"] if n_fft == 512 else _first_occupied(want["occupancy"])
    got_d = np.array([int(w[3]) for w in full])
    got_f = np.array([[float(x) for x in w[7:11]] for w in full])
    assert np.array_equal(got_d, wd) and np.array_equal(wd, picks)
    assert (np.abs(got_f - want["features"]) / np.abs(want["features"])).max() < 1e-5
    assert all(float(w[5]) == TX[int(d)] for w, d in zip(full, wd))


@pytest.mark.parametrize("args,L", [
    (["-m", "scan", "-n", "1024", "-c", "2", "-a", "0"], 512),
    (["-m", "welch", "-n", "2048", "-k", "4", "-b", "2"], 364),
    (["-m", "energy", "-n", "4096", "-k", "3", "-t", "8", "-a", "0"], 4096),
    (["-m", "ref", "-n", "2048", "-k", "12", "-b", "3"], 1000),
], ids=["scan synchronous", "welch batched", "energy 4096 K=3", "ref 2048 K=12 batched"])
def test_engine_option_combinations_against_the_oracle(built, tmp_path, args, L):
    """Less usual combinations of the ce_args — synchronous scan with calibration, batched Welch, short epochs, long epochs at another
    size — each against the oracle run with the configuration the options stand for."""
    opt = dict(zip(args[::2], args[1::2]))
    n, mode = int(opt["-n"]), opt["-m"]
    K = int(opt.get("-k", 8 if mode == "scan" else 10))
    lam = float(opt.get("-t", 4))
    if mode == "scan":
        cfg = cs.cfg_welch_scaled(n, K, lam)          # traffic on the reference's channels
    elif mode == "welch":
        cfg = cs.cfg_welch_scaled(n, K, lam)
    elif mode == "energy":
        cfg = cs.cfg_energy_scaled(n, lam)
        cfg.frames_per_epoch = K
    else:
        cfg = cs.cfg_reference_scaled(n)
        cfg.frames_per_epoch = K
    n_epochs = 9
    if cfg.hop != cfg.fft_len:
        iq, picks, P = _welch_stream(cfg, n_epochs, L, seed=21)
        stride, spf = P * L, n
    else:
        iq, picks = signals.make_epochs(cfg, n_epochs, seed=22, L=L)
        stride, spf = 0, L
    out = _run_harness("engine_harness", ["IQ", str(L), "-g", "0", "-v", "0"] + args, tmp_path, iq)
    _, allp = _epoch_lines(out)
    calib = int(opt.get("-c", 0))
    assert [int(w[1]) for w in allp] == list(range(n_epochs - calib))
    got = np.array([int(w[3]) for w in allp])
    if mode == "ref":
        want = orc.run(cfg, iq, n_epochs, L=spf, epoch_stride=stride)["decision"]
    elif mode == "scan":
        want = picks            # (thresholds come from the engine's own calibration: the driven channel is what must be found)
    else:
        want = _first_occupied(orc.run(cfg, iq, n_epochs, L=spf, epoch_stride=stride)["occupancy"])
    assert np.array_equal(got, np.asarray(want)[calib:]), (got, want)
