class MidpointVI(object):
    pass


class BatchMidpointVI(object):
    pass
