"""Sectioned binary container reader — mirror of BinFileUtils::BinFile
(reference src/binfile_utils.hpp:10-52, src/binfile_utils.cpp:14-140): same method names,
same error texts (thrown by value: quirk Q1 of SURVEY §A.4 is fixed, not replicated)."""
import struct


class BinFile:
    def __init__(self, data, type_, max_version):
        # binfile_utils.cpp:14-62
        if isinstance(data, (str, bytes)) and not isinstance(data, bytes):
            with open(data, "rb") as f:
                data = f.read()
        self.data = memoryview(data)
        ftype = bytes(self.data[:4]).decode("latin1")
        if ftype != type_:
            raise ValueError("Invalid file type. It should be %s and it us %s" % (type_, ftype))
        self.pos = 4
        self.version = self.readU32LE()
        if self.version > max_version:
            raise ValueError("Invalid version. It should be <=%d and it us %d" % (max_version, self.version))
        nsections = self.readU32LE()
        self.sections = {}
        for _ in range(nsections):
            stype = self.readU32LE()
            ssize = self.readU64LE()
            self.sections.setdefault(stype, []).append((self.pos, ssize))
            self.pos += ssize
        self.pos = 0
        self.reading = None

    def _sec(self, section_id, section_pos):
        if section_id not in self.sections:
            raise IndexError("Section does not exist: %d" % section_id)
        lst = self.sections[section_id]
        if section_pos >= len(lst):
            raise IndexError("Section pos too big. There are %d and it's trying to access section: %d" % (len(lst), section_pos))
        return lst[section_pos]

    def startReadSection(self, section_id, section_pos=0):
        start, size = self._sec(section_id, section_pos)
        if self.reading is not None:
            raise IndexError("Already reading a section")
        self.pos = start
        self.reading = (start, size)

    def endReadSection(self, check=True):
        if check and self.pos - self.reading[0] != self.reading[1]:
            raise IndexError("Invalid section size")
        self.reading = None

    def getSectionData(self, section_id, section_pos=0):
        start, size = self._sec(section_id, section_pos)
        return self.data[start:start + size]

    def getSectionSize(self, section_id, section_pos=0):
        return self._sec(section_id, section_pos)[1]

    def readU32LE(self):
        v = struct.unpack_from("<I", self.data, self.pos)[0]
        self.pos += 4
        return v

    def readU64LE(self):
        v = struct.unpack_from("<Q", self.data, self.pos)[0]
        self.pos += 8
        return v

    def read(self, n):
        v = self.data[self.pos:self.pos + n]
        self.pos += n
        return v


def open_existing(filename_or_bytes, type_, max_version):
    """BinFileUtils::openExisting (binfile_utils.cpp:142-144)."""
    if isinstance(filename_or_bytes, str):
        with open(filename_or_bytes, "rb") as f:
            filename_or_bytes = f.read()
    return BinFile(filename_or_bytes, type_, max_version)
