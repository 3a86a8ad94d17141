"""SimpleQueue -- the bounded decode window of utils/queue.py:16-38 (host-side bookkeeping only)."""
import collections


class SimpleQueue(object):
    def __init__(self, maxLen):
        self.maxLen = maxLen
        self.content = collections.deque()
        self.len = 0

    def clear(self):
        self.content = collections.deque()
        self.len = 0

    def full(self):
        return self.len == self.maxLen

    def add(self, item):
        if self.full():
            self.content.popleft()          # drop-oldest; `len` stays at maxLen
        else:
            self.len += 1
        self.content.append(item)

    def get_all(self):
        return list(self.content)
