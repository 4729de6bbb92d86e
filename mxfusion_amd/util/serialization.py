"""Checkpoint helpers (mxfusion/util/serialization.py:26-135): the same zip layout and file names as the reference's
Inference.save (inference/inference.py:255-310)."""
import io
import json
import zipfile

import numpy as np

from ..common.exceptions import SerializationError

__GRAPH_JSON_VERSION__ = '1.0'
SERIALIZATION_VERSION = '2.0'
DEFAULT_ZIP = 'inference.zip'
FILENAMES = {
    'graphs': 'graphs.json',
    'mxnet_params': 'mxnet_parameters.npz',
    'mxnet_constants': 'mxnet_constants.npz',
    'variable_constants': 'variable_constants.json',
    'configuration': 'configuration.json',
    'version_file': 'version.json',
}


def make_numpy(obj):
    """serialization.py:94-113: {key: device tensor | ndarray} -> {key: ndarray}."""
    if not isinstance(obj, dict):
        raise SerializationError("make_numpy expects a dictionary of arrays")
    out = {}
    for k, v in obj.items():
        if isinstance(v, np.ndarray):
            out[k] = v
        elif hasattr(v, 'detach'):
            out[k] = v.detach().cpu().numpy()
        else:
            raise SerializationError("make_numpy expects a dictionary of arrays")
    return out


def load_json_from_zip(zip_filename, target_file):
    with zipfile.ZipFile(zip_filename, 'r') as zf:
        return json.load(io.StringIO(zf.read(target_file).decode()))


def load_parameters(npz_filename, zip_file):
    """serialization.py:115-135.  A missing member or an archive without arrays loads as {} (the reference writes an empty .npz when there
    are no constants); a member that IS there but cannot be read -- truncated, corrupt, not an .npz -- raises SerializationError instead of
    silently restoring nothing."""
    if npz_filename not in zip_file.namelist():
        return {}
    raw = zip_file.read(npz_filename)
    if len(raw) == 0:
        return {}
    try:
        loaded = np.load(io.BytesIO(raw))
        return {k: loaded[k] for k in loaded.files}
    except Exception as e:      # zipfile.BadZipFile, OSError, ValueError, pickle errors ...
        raise SerializationError('cannot read %s from the checkpoint: %s' % (npz_filename, e))


def write_zip(zip_filename, json_files, npz_files):
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, 'a', zipfile.ZIP_DEFLATED, False) as zf:
        for name, obj in json_files.items():
            zf.writestr(name, json.dumps(obj, ensure_ascii=False))
        for name, arrays in npz_files.items():
            b = io.BytesIO()
            np.savez(b, **make_numpy(arrays))
            zf.writestr(name, b.getvalue())
    with open(zip_filename, 'wb') as f:
        f.write(buf.getvalue())
