"""``get_monitor`` of the reference (rpo/utils/monitor.py:3-27): a root ``logging`` logger that echoes the launching
script (and optional package files) so a run's console log records its own configuration.  Cosmetic; kept because
scripts/evopf_exp_sac.py:8-9 imports it."""
import logging


def get_monitor(logpath, filepath, package_files=(), displaying=True, saving=False, debug=False):
    log = logging.getLogger()
    level = logging.DEBUG if debug else logging.INFO
    log.setLevel(level)
    handlers = []
    if saving:
        handlers.append(logging.FileHandler(logpath, mode="a"))
    if displaying:
        handlers.append(logging.StreamHandler())
    for h in handlers:
        h.setLevel(level)
        log.addHandler(h)
    for path in (filepath,) + tuple(package_files):
        log.info(path)
        with open(path, "r") as f:
            log.info(f.read())
    return log
