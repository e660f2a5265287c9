class ImageEncoder(object):
    pass
