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
