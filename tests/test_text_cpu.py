"""Host logic of the paragraph query grammar (no GPU): the cases of nidx_paragraph/src/query_parser/tokenizer.rs's own
tests (test_empty_query, test_simple_query, quotes / exclusions / unclosed quotes) and query_parser.rs's stop-word test
shape."""
from nucliadb_amd.text import facet_ancestors, is_valid_facet, parse_query


def lit(*words):
    return [("literal", w) for w in words]


def test_empty_queries():
    for q in ("", "    ", "  - - -   - - -  ", '"  "', "!@#~&/()=?"):
        assert parse_query(q) == [], q


def test_literals_quotes_and_exclusions():
    assert parse_query("This is a simple query") == lit("this", "is", "a", "simple", "query")
    assert parse_query('some "exact match" here') == [("literal", "some"), ("quoted", "exact match"), ("literal", "here")]
    assert parse_query("keep -drop this") == [("literal", "keep"), ("excluded", "drop"), ("literal", "this")]
    # punctuation inside a literal splits it; the dash inside a word is not an exclusion
    assert parse_query("do-stuff a.b") == lit("do", "stuff", "a", "b")
    # an unclosed quote is dropped, the words stay
    assert parse_query('shoupd + enaugh"') == lit("shoupd", "enaugh")
    assert parse_query('"unclosed words') == lit("unclosed", "words")
    assert parse_query('"Quoted, With! Punctuation"') == [("quoted", "quoted with punctuation")]
    assert parse_query("some ' document") == lit("some", "document")


def test_stop_words_keep_the_last_token():
    stop = {"is", "a", "for", "the", "and"}
    assert parse_query("nuclia is a database for unstructured data", stop) == lit("nuclia", "database", "unstructured", "data")
    assert parse_query("nuclia is a database for the", stop) == lit("nuclia", "database", "the")
    assert parse_query("is a for and", stop) == lit("and")
    assert parse_query('nuclia "is" a database', stop) == [("literal", "nuclia"), ("quoted", "is"), ("literal", "database")]


def test_facet_paths():
    assert facet_ancestors("/l/labelset/label") == ["/l", "/l/labelset", "/l/labelset/label"]
    assert is_valid_facet("/l") and is_valid_facet("/e/PERSON") and not is_valid_facet("") and not is_valid_facet("l/x")


def test_adapt_text_quote_fixing():
    """nidx_text/tests/test_search.rs:291-310 (test_quote_fixing) on TextReaderService::adapt_text (reader.rs:357-365)."""
    from nucliadb_amd.text import TextSearcher

    for body in ('"enough test"', 'enough test"', '"enough test'):
        assert TextSearcher.adapt_text(body) == '"enough test"'
    assert TextSearcher.adapt_text("") == "" and TextSearcher.adapt_text("enough test") == "enough test"
    assert TextSearcher.adapt_text('a "b c" d "e') == '"a b c d e"'
