"""Text log with the reference's line format (logger/logger.py:103-137, 217-232):
``04OCTOBER2026-10:07:33:: INFO   : message`` -- written to a file and echoed when verbose."""
import errno
import os
import time

VALID_MTYPES = ("INFO", "ERROR", "WARNING")
_PAD = max(len(m) for m in VALID_MTYPES)


class Entry:
    def __init__(self, mtype, entry, timestamp=True):
        if not isinstance(mtype, str):
            raise TypeError("mtype must be a str")
        if not isinstance(entry, str):
            raise TypeError("entry must be a str")
        if mtype.upper() not in VALID_MTYPES:
            raise TypeError("mtype must be one of 'INFO', 'ERROR' or 'WARNING'")
        self.rtime = time.time()
        self.mtime = time.localtime()
        self.mtype = mtype
        self.message = entry
        self.timestamp = timestamp

    def time_str(self, fmt='%d%B%Y-%H:%M:%S'):
        return time.strftime(fmt, self.mtime).upper()

    def __repr__(self):
        return "Entry(mtype={!r}, entry={!r}, timestamp={})".format(self.mtype, self.message,
                                                                   self.timestamp)

    def __str__(self):
        head = ':: '.join([self.time_str(), format(self.mtype, str(_PAD))])
        if not self.timestamp:
            head = ' ' * len(head)
        lines = self.message.split('\n')
        indent = ' ' * (len(head) + 2)
        body = '\n'.join([lines[0]] + [indent + ln for ln in lines[1:]])
        return ': '.join([head, body])


class Log:
    def __init__(self, fname, verbose=True):
        self._entries = {}
        self._filename = fname
        self.verbose = verbose

    @classmethod
    def combine_logs(cls, log1, log2, filename, delete_old_logs):
        """Merge two logs into a new, time-sorted one (logger/logger.py:17-62)."""
        for old in (log1.filename, log2.filename):
            if (delete_old_logs or filename == old) and os.path.exists(old):
                os.remove(old)
        merged = sorted(list(log1.entries.values()) + list(log2.entries.values()),
                        key=lambda e: e.rtime)
        new = cls(filename, verbose=log1.verbose or log2.verbose)
        new.entries = dict(enumerate(merged))
        for e in merged:
            new.write_entry(e)
        return new

    @property
    def filename(self):
        return self._filename

    @property
    def entries(self):
        return self._entries

    @entries.setter
    def entries(self, new_entries):
        self._entries = new_entries

    def __str__(self):
        return '\n'.join(str(self.entries[k]) for k in sorted(self.entries))

    def add_entry(self, mtype, entry, timestamp=True):
        dcy = os.path.dirname(self.filename)
        if not os.path.exists(dcy):
            raise FileNotFoundError(errno.ENOTDIR, os.strerror(errno.ENOTDIR), dcy)
        new = Entry(mtype, entry, timestamp)
        self._entries[len(self._entries) + 1] = new
        if self.verbose:
            print(new)
        self.write_entry(new)

    def write_entry(self, entry):
        nonempty = os.path.exists(self.filename) and os.path.getsize(self.filename) > 0
        with open(self.filename, 'at+') as f:
            f.write(('\n' if nonempty else '') + str(entry))
