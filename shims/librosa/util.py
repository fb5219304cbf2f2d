import os


def find_files(directory, ext=("wav",), recurse=True, **_kw):
    ext = (ext,) if isinstance(ext, str) else tuple(ext)
    out = []
    for root, _dirs, files in os.walk(os.path.abspath(os.path.expanduser(directory))):
        for f in files:
            if f.rsplit(".", 1)[-1].lower() in ext:
                out.append(os.path.join(root, f))
        if not recurse:
            break
    return sorted(out)
