"""Buffered logger with the reference's interface (``pyrhe/src/util/logger.py:3-25``):
``_log`` buffers + prints, ``_debug`` prints in debug mode, ``_save_log`` writes the buffer once."""


class Logger:
    def __init__(self, output_file=None, suppress=False, debug_mode=True):
        self.msgs = []
        self.output_file = output_file
        self.suppress = suppress
        self.debug_mode = debug_mode

    def _debug(self, msg):
        if self.debug_mode:
            print(msg)

    def _log(self, *args, end="\n"):
        line = " ".join(str(a) for a in args) + end
        self.msgs.append(line)
        if not self.suppress:
            print(line, end="")

    def _save_log(self):
        if self.output_file is not None:
            with open(self.output_file, "w") as fd:
                fd.writelines(self.msgs)
