"""SequenceFiles: the on-disk sequence format opened without a GPU."""
import numpy as np
import pytest

from evreal_amd import synth
from evreal_amd.dataset import SequenceFiles


def test_sequence_files_with_and_without_frames(tmp_path):
    w = synth.write_sequence(str(tmp_path / 'a'), 1, 5000, 1.0e5, 32, 24, 100.0)
    s = SequenceFiles.open(str(tmp_path / 'a'))
    assert s.num_events == 5000 and s.images is not None and s.images.shape[1:] == (24, 32, 1)
    assert np.array_equal(np.asarray(s.t), w['t']) and np.array_equal(np.asarray(s.xy), w['xy'])
    assert s.metadata_resolution() == [24, 32]
    assert np.array_equal(s.image_event_indices, w['image_event_indices'])
    synth.write_sequence(str(tmp_path / 'b'), 2, 1000, 1.0e5, 32, 24, with_images=False)
    s = SequenceFiles.open(str(tmp_path / 'b'))
    assert s.images is None and s.frame_stamps is None and s.num_events == 1000


def test_sequence_files_length_mismatch(tmp_path):
    synth.write_sequence(str(tmp_path / 'c'), 3, 1000, 1.0e5, 32, 24, with_images=False)
    np.save(tmp_path / 'c' / 'events_p.npy', np.zeros(999, np.uint8))
    with pytest.raises(AssertionError, match='do not match'):
        SequenceFiles.open(str(tmp_path / 'c'))


def test_window_tables_and_cost_without_a_gpu(tmp_path):
    """MemMapDataset needs no GPU until upload(): the window tables (dataset.py:104-130,168-186) and the rank-assignment
    weight of SURVEY 8e (windows x padded pixels) are host work."""
    from evreal_amd.dataset import MemMapDataset
    synth.write_sequence(str(tmp_path / 'a'), 1, 5000, 1.0e5, 32, 24, 100.0)
    ds = MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method={'method': 'k_events', 'k': 500, 'sliding_window_w': 0})
    assert len(ds) == 10 and ds.sensor_resolution == [24, 32]
    tb = ds.table()
    assert tb['idx0'].tolist() == list(range(0, 5000, 500)) and tb['valid'].all()
    assert ds.window_cost() == 10 * 32 * 32                       # 24 x 32 pads to 32 x 32
    ds2 = MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method={'method': 'between_frames'})
    assert ds2.window_cost() == (ds2.num_frames - 1) * 32 * 32


def test_zero_reference_frames_fall_through(tmp_path):
    """images.npy / images_ts.npy with zero frames: the reference's `[ts.item() for ts in frame_stamps]` yields [] and the
    resolution comes from max(xy) + 1 (dataset.py:262,270-281)."""
    from evreal_amd.dataset import MemMapDataset
    w = synth.write_sequence(str(tmp_path / 'z'), 4, 2000, 1.0e5, 32, 24, with_images=False)
    np.save(tmp_path / 'z' / 'images.npy', np.zeros((0, 24, 32, 1), np.uint8))
    np.save(tmp_path / 'z' / 'images_ts.npy', np.zeros((0, 1), np.float64))
    np.save(tmp_path / 'z' / 'image_event_indices.npy', np.zeros((0, 1), np.int64))
    import os
    if os.path.exists(tmp_path / 'z' / 'metadata.json'):
        os.remove(tmp_path / 'z' / 'metadata.json')
    ds = MemMapDataset(str(tmp_path / 'z'), num_bins=5, voxel_method={'method': 'k_events', 'k': 500, 'sliding_window_w': 0})
    assert ds.num_frames == 0 and ds.frame_ts == []
    assert ds.sensor_resolution == [int(w['xy'][:, 1].max()) + 1, int(w['xy'][:, 0].max()) + 1]


def test_out_of_sensor_coordinates_raise_before_upload(tmp_path):
    """A sensor_resolution smaller than the coordinates: the reference raises from index_put_ on the first such window
    (SURVEY 8a quirk 6); here the sequence is refused before anything reaches the GPU."""
    from evreal_amd.dataset import MemMapDataset
    synth.write_sequence(str(tmp_path / 'a'), 1, 5000, 1.0e5, 32, 24, 100.0)
    ds = MemMapDataset(str(tmp_path / 'a'), sensor_resolution=[20, 32], num_bins=5,
                       voxel_method={'method': 'k_events', 'k': 500, 'sliding_window_w': 0})
    with pytest.raises(IndexError, match='outside the 32x20 sensor'):
        ds.upload()


def test_validation_covers_the_windows_events_only(tmp_path):
    """the reference fails from index_put_ only on a window it actually voxelises; events no window of this dataset object
    touches (here: beyond max_length) must not refuse the sequence.  The polarity column is checked in one pass, and the validated
    columns are handed out once more from the cache (eval's grouping validates before the batch that uploads is formed)."""
    from evreal_amd.dataset import MemMapDataset
    w = synth.write_sequence(str(tmp_path / 'a'), 1, 5000, 1.0e5, 32, 24, 100.0)
    xy = w['xy'].copy()
    xy[4000:, 0] = 40                                             # beyond the 32-wide sensor, in the last 1000 events only
    np.save(tmp_path / 'a' / 'events_xy.npy', xy)
    vm = {'method': 'k_events', 'k': 500, 'sliding_window_w': 0}
    with pytest.raises(IndexError, match='outside the 32x24 sensor'):
        MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method=vm).host_events()
    ds = MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method=vm, max_length=7)      # length = min(L, max_length + 1) (dataset.py:190-191): windows cover events [0, 4000)
    assert len(ds) == 8
    cols = ds.host_events(keep=True)
    assert cols[0].dtype == np.int16 and cols[1].dtype == np.float64 and cols[2].dtype == np.uint8 and len(cols[1]) == 5000
    assert ds.host_events() is cols and ds._host_cols is None     # handed over once, then released
    np.save(tmp_path / 'a' / 'events_p.npy', (w['p'].astype(np.int8) * 2 - 1))              # -1 / +1 instead of 0 / 1
    with pytest.raises(ValueError, match='must hold 0/1'):
        MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method=vm, max_length=7).host_events()
    np.save(tmp_path / 'a' / 'events_p.npy', np.full(5000, 2, np.uint8))
    with pytest.raises(ValueError, match='must hold 0/1'):
        MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method=vm, max_length=7).host_events()
    # a float file holding exactly 0.0 / 1.0 is what the reference's p.astype(float32)*2-1 (dataset.py:227) accepts: so do we
    np.save(tmp_path / 'a' / 'events_p.npy', w['p'].astype(np.float32))
    cols = MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method=vm, max_length=7).host_events()
    assert cols[2].dtype == np.uint8 and np.array_equal(cols[2], w['p'].astype(np.uint8))
    np.save(tmp_path / 'a' / 'events_p.npy', w['p'].astype(np.float32) * 0.5)                # 0.0 / 0.5: not polarities
    with pytest.raises(ValueError, match='must hold 0/1'):
        MemMapDataset(str(tmp_path / 'a'), num_bins=5, voxel_method=vm, max_length=7).host_events()
