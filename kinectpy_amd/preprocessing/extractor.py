"""Drop-in for KinectPy's preprocessing/extractor.py (reference lines 17-194).

The reference obtains the per-pixel XYZ `.dat` files by shelling out to an external Windows binary
(`offline_processor.exe <mkv> --gpu --pointcloud`, extractor.py:68-80).  MKV demuxing and body
tracking stay out of scope (proprietary SDK); the depth -> XYZ step that binary performs is done here
on the MI355X (kpx_unproject_u16).  `frame_source(mkv_fp)` supplies the demuxed frames.
"""
import logging
import os
from pathlib import Path
from typing import Callable, Iterable, List, Optional, Tuple

import numpy as np

from .. import ops

FrameSource = Callable[[str], Tuple[np.ndarray, Iterable[Tuple[int, np.ndarray]]]]


class MKVFilesProcessing(object):
    def __init__(self, mkv_fps: List[str] = [], output_dirs: List[str] = [], offline_processor_fp: str = None,
                 number_of_joints: int = 32, frame_source: Optional[FrameSource] = None) -> None:
        """Same arguments as the reference (extractor.py:18-24) plus `frame_source`: a callable
        mkv_fp -> (xy_table f32 (H*W,2), iterable of (timestamp, depth u16 (H*W,))) replacing the MKV
        demux of the external binary."""
        self.mkv_fps, self.output_dirs = mkv_fps, output_dirs
        self.offline_processor_fp, self.number_of_joints = offline_processor_fp, number_of_joints
        self.frame_source = frame_source
        self._verify_setup(mkv_fps, output_dirs)
        for output_dir in output_dirs:
            self._create_folder_structure(output_dir)

    def extract(self, color: bool = False, depth: bool = False, skeleton: bool = False, pointcloud: bool = False,
                batch: int = 64) -> None:
        """extractor.py:48-86.  pointcloud=True writes `<ts>_depth.dat` (int16 (H*W,3) XYZ mm) under
        <output_dir>/depths -- the layout utils/io.py:15-20 reads.  color / skeleton need the MKV colour
        track and the body-tracking SDK: out of scope; depth raises as in the reference."""
        if depth:
            raise NotImplementedError('Currently depth images cannot be extracted')       # extractor.py:75-77
        if color or skeleton:
            raise NotImplementedError('colour / skeleton tracks need the Azure Kinect SDK (out of scope)')
        if not pointcloud:
            return
        for mkv_fp, output_dir in zip(self.mkv_fps, self.output_dirs):
            logging.info(f'Starting to process MKV File {mkv_fp}\nIt will be saved under directory: {output_dir}\n')
            xy, frames = self.frame_source(mkv_fp)
            stamps, buf = [], []

            def flush():
                if not buf:
                    return
                xyz = ops.unproject_u16(np.stack(buf).reshape(-1), xy, len(buf)).cpu().numpy()
                for ts, a in zip(stamps, xyz):
                    if f'{ts}_depth.dat'.startswith('0_'):  # "0_*" files (timestamp exactly 0) are empty frames (extractor.py:150-154)
                        continue
                    a.tofile(os.path.join(output_dir, 'depths', f'{ts}_depth.dat'))
                stamps.clear(); buf.clear()

            for ts, d in frames:
                stamps.append(ts); buf.append(np.asarray(d, dtype=np.uint16).reshape(-1))
                if len(buf) == batch:
                    flush()
            flush()
        logging.info('Extraction is done!')

    def align_skeletons(self):
        """extractor.py:89-126: registered = (F,J,3) @ inv(R) + t with the saved sub->master transform
        (the master itself uses transformation_master_sub_1.npy, as the reference does)."""
        import pandas as pd
        for output_dir in self.output_dirs:
            synced = pd.read_csv(os.path.join(output_dir, 'skeleton', 'synced_positions_3d.csv'), index_col='timestamp')
            device = os.path.basename(os.path.normpath(output_dir))
            name = f'transformation_master_{device}.npy' if device != 'master_1' else 'transformation_master_sub_1.npy'
            T = np.load(os.path.join(self.output_dirs[0], name))
            vals = transform_joint_rows(synced.values, T, self.number_of_joints)
            pd.DataFrame(columns=synced.columns, data=vals, index=synced.index).to_csv(
                os.path.join(output_dir, 'skeleton', 'registered_positions_3d.csv'))

    def _create_folder_structure(self, output_dir):
        for sub in ('color', 'filtered_pointclouds', 'filtered_and_registered_pointclouds', 'pointclouds', 'skeleton', 'depths'):
            Path(os.path.join(output_dir, sub)).mkdir(parents=True, exist_ok=True)

    def _verify_setup(self, mkv_fps, output_dirs):
        """extractor.py:167-181: same exceptions; the external binary is only required when no
        frame_source replaces it."""
        if self.frame_source is None and not (self.offline_processor_fp and os.path.isfile(self.offline_processor_fp)):
            raise FileNotFoundError('Make sure that the offline_processor.exe path is correct')
        if len(mkv_fps) != len(output_dirs) or len(mkv_fps) == 0:
            raise Exception('Make sure to give two lists, where each mkv file'
                            'has a correspondent directory to be output')


def transform_joint_rows(values: np.ndarray, transformation: np.ndarray, number_of_joints: Optional[int] = None) -> np.ndarray:
    """(F, 3J) joint rows -> x @ inv(R) + t on the device (utils/processing.py:357-383 transform_joints,
    extractor.py:109-122)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    j = number_of_joints if number_of_joints is not None else v.shape[1] // 3
    T = np.asarray(transformation, dtype=np.float64)
    out = ops.joints_affine(v.reshape(-1, 3), np.linalg.inv(T[:3, :3]), T[:3, 3])
    return out.cpu().numpy().reshape(v.shape[0], j * 3)
